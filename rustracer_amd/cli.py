"""Host mirror of src/cli.rs + src/main.rs: the `rustracer` command line over the MI355X render core.

    python -m rustracer_amd convert  HDR LDR [-f FACTOR] [-g GAMMA] [-v]
    python -m rustracer_amd demo     OUTPUT  [--width W --height H --angle-deg A -a ALGORITHM ...]
    python -m rustracer_amd render   INPUT OUTPUT [same flags as demo, minus --orthogonal]

Same subcommands, flags, short options and defaults as the reference (src/cli.rs:11-51, :55-443);
`completion` (shell completion files) is not mirrored.  The only call that differs from the
reference is ImageTracer.fire_all_rays, which goes through the C-ABI to the GPU.
"""
import argparse
import os
import sys

import numpy as np

from .camera import OrthogonalCamera, PerspectiveCamera
from .hdrimage import HdrImage, HdrImageErr
from .imagetracer import ImageTracer
from .material import BLACK, WHITE
from .render import DummyRenderer, FlatRenderer, OnOffRenderer, PathTracer
from .scene import Cli, Scene, SceneErr, parse_f32
from .scenes import demo_world
from .transformation import rotation_z, to_radians, translation

F = np.float32
# defaults of src/cli.rs:11-51
FACTOR, GAMMA, WIDTH, HEIGHT, ANGLE_DEG = "1.0", "1.0", "640", "480", "0.0"
ALGORITHM, NUM_OF_RAYS, MAX_DEPTH, INIT_STATE, INIT_SEQ, ANTI_ALIASING = "pathtracer", "10", "3", "42", "54", "1"
RUSSIAN_ROULETTE_LIMIT = 3          # hard-coded in src/main.rs:218, :312


class CliErr(Exception):
    """ConvertErr / DemoErr / RenderErr of src/error.rs: message only."""


def _tone_flags(p):
    p.add_argument("-f", "--factor", metavar="FACTOR", default=FACTOR, help="Normalization factor")
    p.add_argument("-g", "--gamma", metavar="GAMMA", default=GAMMA, help="Gamma transfer function parameter")


def _render_flags(p):
    p.add_argument("-v", "--verbose", action="store_true", help="Print stdout information")
    p.add_argument("--output-pfm", action="store_true", help="Output also hdr image")
    p.add_argument("--width", metavar="WIDTH", default=WIDTH, help="Image width")
    p.add_argument("--height", metavar="HEIGHT", default=HEIGHT, help="Image height")
    p.add_argument("--angle-deg", metavar="ANGLE_DEG", default=ANGLE_DEG, help="View angle (in degrees)")
    _tone_flags(p)
    p.add_argument("-a", "--algorithm", metavar="ALGORITHM", default=ALGORITHM,
                   choices=["onoff", "flat", "pathtracer"], help="Rendering algorithm")
    p.add_argument("-n", "--num-of-rays", metavar="NUM_OF_RAYS", default=NUM_OF_RAYS, help="Number of rays")
    p.add_argument("-m", "--max-depth", metavar="MAX_DEPTH", default=MAX_DEPTH, help="Maximum depth")
    p.add_argument("--init-state", metavar="INIT_STATE", default=INIT_STATE, help="Initial random seed (positive number)")
    p.add_argument("--init-seq", metavar="INIT_SEQ", default=INIT_SEQ, help="Identifier of the random sequence (positive number)")
    p.add_argument("--anti-aliasing", metavar="ANTI_ALIASING", default=ANTI_ALIASING, help="Anti-aliasing level")


def build_cli():
    """src/cli.rs:55-443."""
    ap = argparse.ArgumentParser(prog="rustracer", description="a multi-threaded raytracer in pure rust (MI355X render core)")
    sub = ap.add_subparsers(dest="command", required=True)
    c = sub.add_parser("convert", help="Convert HDR (pfm) image to LDR (ff|png) image")
    c.add_argument("HDR", help="Input pfm image")
    c.add_argument("LDR", help="Output image [possible formats: ff, png]")
    c.add_argument("-v", "--verbose", action="store_true", help="Print stdout information")
    _tone_flags(c)
    d = sub.add_parser("demo", help="Render a demo scene (hard-coded in main)")
    d.add_argument("OUTPUT", help="Output image [possible formats: ff, png]")
    d.add_argument("--orthogonal", action="store_true", help="Use orthogonal camera instead of perspective camera")
    _render_flags(d)
    r = sub.add_parser("render", help="Render a scene from file (yaml formatted)")
    r.add_argument("INPUT", help="Input scene file")
    r.add_argument("OUTPUT", help="Output image [possible formats: ff, png]")
    _render_flags(r)
    return ap


def _f32(text, what):
    try:
        return float(parse_f32(text))
    except ValueError:
        raise CliErr(f"invalid floating-point number for `{what}`: {text!r}")


def _uint(text, what, bits):
    if not (text.isascii() and text.lstrip("+").isdigit() and text.count("+") <= 1) or int(text) >= 1 << bits:
        raise CliErr(f"invalid integer for `{what}`: {text!r}")
    return int(text)


def _check_ldr(path):
    """`check!` of src/misc.rs:67-78: only png and farbfeld outputs."""
    ext = os.path.splitext(path)[1].lstrip(".").lower()
    if ext not in ("png", "ff"):
        raise CliErr(f"unsupported ldr file format {ext!r}")


def _info(args, msg):
    if args.verbose:
        print(f"[info] {msg}")


def convert(args):
    """src/main.rs:78-107."""
    factor, gamma = _f32(args.factor, "factor"), _f32(args.gamma, "gamma")
    try:
        img = HdrImage.read_pfm_file(args.HDR)
    except (HdrImageErr, OSError) as e:
        raise CliErr(str(e))
    _check_ldr(args.LDR)
    _info(args, f"{args.HDR!r} has been read from disk")
    img.normalize_image(factor)
    img.clamp_image()
    img.write_ldr_file(args.LDR, gamma)
    _info(args, f"{args.LDR!r} has been written to disk")


def _numbers(args):
    return dict(factor=_f32(args.factor, "factor"), gamma=_f32(args.gamma, "gamma"),
                width=_uint(args.width, "width", 32), height=_uint(args.height, "height", 32),
                angle_deg=_f32(args.angle_deg, "angle-deg"), num_of_rays=_uint(args.num_of_rays, "num-of-rays", 32),
                max_depth=_uint(args.max_depth, "max-depth", 32), init_state=_uint(args.init_state, "init-state", 64),
                init_seq=_uint(args.init_seq, "init-seq", 64), aa=_uint(args.anti_aliasing, "anti-aliasing", 32))


def _renderer(args, world, n):
    """src/main.rs:214-222, :308-316."""
    if args.algorithm == "onoff":
        return OnOffRenderer(world, BLACK, WHITE)
    if args.algorithm == "flat":
        return FlatRenderer(world, BLACK)
    if args.algorithm == "pathtracer":
        return PathTracer(world, BLACK, n["num_of_rays"], n["max_depth"], RUSSIAN_ROULETTE_LIMIT)
    return DummyRenderer()


def pfm_path_for(ldr_file):
    """ldr_file.with_extension("").with_extension("pfm") (src/main.rs:225, :319): one extension is stripped, then
    whatever extension is left is REPLACED: "a.b.png" -> "a.pfm"."""
    return os.path.splitext(os.path.splitext(ldr_file)[0])[0] + ".pfm"


def _trace_and_write(args, n, world, camera):
    """The common tail of demo and render (src/main.rs:223-248, :317-342)."""
    img = HdrImage(n["width"], n["height"])
    tracer = ImageTracer(img, camera)
    tracer.fire_all_rays(_renderer(args, world, n), n["init_state"], n["init_seq"], n["aa"])
    if args.output_pfm:
        hdr_file = pfm_path_for(args.OUTPUT)
        img.write_pfm_file(hdr_file, "little")
        _info(args, f"{hdr_file!r} has been written to disk")
    img.normalize_image(n["factor"])
    img.clamp_image()
    img.write_ldr_file(args.OUTPUT, n["gamma"])
    _info(args, f"{args.OUTPUT!r} has been written to disk")
    return tracer


def demo(args):
    """src/main.rs:112-250."""
    n = _numbers(args)
    _check_ldr(args.OUTPUT)
    world = demo_world()
    tr = rotation_z(to_radians(F(F(n["angle_deg"]) + F(230.0)))) * translation((-3.5, 0.0, 0.7))
    aspect = float(F(F(n["width"]) / F(n["height"])))
    camera = OrthogonalCamera(aspect, tr) if args.orthogonal else PerspectiveCamera(1.0, aspect, tr)
    _info(args, f"generating an image ({n['width']}, {n['height']})")
    return _trace_and_write(args, n, world, camera)


def render(args):
    """src/main.rs:255-344."""
    n = _numbers(args)
    _check_ldr(args.OUTPUT)
    _info(args, f"reading scene from file {args.INPUT!r}")
    try:
        scene = Scene.read_scene_file(args.INPUT, Cli(float(F(F(n["width"]) / F(n["height"]))), n["angle_deg"]))
    except SceneErr as e:
        raise CliErr(f"{args.INPUT}: {e}")
    _info(args, f"generating an image ({n['width']}, {n['height']})")
    return _trace_and_write(args, n, scene.shapes, scene.camera)


def main(argv=None):
    """src/main.rs:61-73: dispatch; errors go to stderr with exit status 1 (src/misc.rs:46-57)."""
    args = build_cli().parse_args(argv)
    try:
        {"convert": convert, "demo": demo, "render": render}[args.command](args)
    except (CliErr, HdrImageErr) as e:
        print(f"[error] {e}", file=sys.stderr)
        return 1
    return 0
