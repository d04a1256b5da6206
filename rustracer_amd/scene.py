"""Host mirror of src/scene.rs: the scene-file lexer and indentation-sensitive parser.

Runs once per scene on the host and produces the Camera + World that are flattened into the
C-ABI tables (include/rustracer_amd.h).  The grammar, the token stream, the error kinds and the
source locations they carry follow the reference (src/scene.rs:387-1308) so that a file the
reference accepts builds the same scene here and a file it rejects is rejected at the same
line and column.  The lexer (`InputStream`: one-character look-behind, location tracking, comment and
whitespace skipping, the `match_*` / `parse_*` entry points) mirrors the reference's structure method for method,
because reproducing its token locations and error kinds requires the same one-slot `unread` behaviour; it is host
glue outside the GPU path.
"""
import io
import math
from dataclasses import dataclass
from fractions import Fraction

import numpy as np

from .camera import OrthogonalCamera, PerspectiveCamera
from .hdrimage import HdrImage, HdrImageErr
from .material import (BLACK, WHITE, CheckeredPigment, DiffuseBRDF, ImagePigment, Material, SpecularBRDF,
                       UniformPigment)
from .shape import Plane, Sphere
from .transformation import (Transformation, rotation_x, rotation_y, rotation_z, scaling, to_radians,
                             translation)
from .world import World

F = np.float32
SYMBOLS = "\n -:[,]#"          # src/scene.rs:32
KEYWORDS = ("camera", "checkered", "color", "colors", "compose", "diffuse", "distance", "image", "material",
            "materials", "name", "plane", "ratio", "rotationx", "rotationy", "rotationz", "scaling", "shape",
            "shapes", "specular", "sphere", "transformation", "transformations", "translation", "type", "uniform")


@dataclass(frozen=True)
class SourceLocation:
    """src/scene.rs:35-41."""
    line_num: int = 1
    col_num: int = 1


class SceneErr(Exception):
    """src/error.rs SceneErr: `kind` is the variant name, `loc` the SourceLocation it carries."""

    def __init__(self, kind, loc, msg):
        super().__init__(f"<{loc.line_num}:{loc.col_num}> {msg}" if loc else msg)
        self.kind, self.loc, self.msg = kind, loc, msg


@dataclass
class Token:
    """src/scene.rs:79-88; kind in {identifier, keyword, number, stop, string, symbol}."""
    kind: str
    loc: SourceLocation
    value: object = None

    def shown(self):
        if self.kind == "stop":
            return "\x00"
        if self.kind == "number":
            return repr(float(self.value)) if float(self.value) != int(self.value) else str(int(self.value))
        return str(self.value)


@dataclass
class Cli:
    """src/scene.rs `Cli`: the command-line values a scene file may refer to (RATIO) or that
    the parser applies (angle_deg, src/scene.rs:1202)."""
    aspect_ratio: float = 1.0
    angle_deg: float = 0.0


def parse_f32(text):
    """f32::from_str: the binary32 nearest to the decimal literal, ties to even (exact, no
    double rounding through binary64).  Raises ValueError like Rust's ParseFloatError."""
    t = text.strip()
    if not t or t != text:
        raise ValueError(text)
    body = t[1:] if t[0] in "+-" else t
    if not body or not all(c.isdigit() or c in ".eE+-" for c in body):
        raise ValueError(text)
    mant, _, exp = body.replace("E", "e").partition("e")
    if "e" in body.replace("E", "e") and (not exp or not exp.lstrip("+-").isdigit() or exp.count("+") + exp.count("-") > 1):
        raise ValueError(text)
    if mant.count(".") > 1 or not any(c.isdigit() for c in mant) or any(c in "+-" for c in mant):
        raise ValueError(text)
    ip, _, fp = mant.partition(".")
    exact = Fraction(int((ip or "0") + fp), 10 ** len(fp)) * (Fraction(10) ** int(exp or "0"))
    if t[0] == "-":
        exact = -exact
    with np.errstate(over="ignore"):
        approx = F(float(exact)) if abs(exact) < Fraction(2) ** 200 else F(math.copysign(math.inf, exact))
    if not np.isfinite(approx):
        return approx
    best = approx
    for cand in (np.nextafter(approx, F(-np.inf)), np.nextafter(approx, F(np.inf))):
        if not np.isfinite(cand):
            continue
        dc, db = abs(Fraction(float(cand)) - exact), abs(Fraction(float(best)) - exact)
        if dc < db or (dc == db and (int(cand.view(np.uint32)) & 1) == 0 and (int(best.view(np.uint32)) & 1) == 1):
            best = cand
    if t[0] == "-" and best == 0:
        best = F(-0.0)
    return best


def _f32_to_u32_sat(x):
    """Rust `as u32` (src/scene.rs:772): saturating, NaN -> 0."""
    x = float(x)
    if not x > 0.0:
        return 0
    return 0xFFFFFFFF if x >= 4294967296.0 else int(x)


class Var:
    """src/scene.rs:1311-1346: named values visible to later blocks."""

    def __init__(self):
        self.colors = {"BLACK": BLACK, "WHITE": WHITE}
        self.materials = {}
        self.transformations = {"IDENTITY": Transformation()}
        self.vectors = {"E1": (1.0, 0.0, 0.0), "E2": (0.0, 1.0, 0.0), "E3": (0.0, 0.0, 1.0)}


class Scene:
    """src/scene.rs:1349-1372."""

    def __init__(self):
        self.camera = None
        self.shapes = None

    @staticmethod
    def read_scene_file(path, cli):
        try:
            with open(path, "rb") as f:
                data = f.read()
        except OSError as e:
            raise SceneErr("SceneFileReadFailure", None, f"scene file read failure: {e}")
        return InputStream(io.BytesIO(data)).parse_scene(cli)


class InputStream:
    """src/scene.rs:120-1308: one-character / one-token look-ahead reader + recursive-descent parser."""

    def __init__(self, reader):
        if isinstance(reader, str):
            reader = io.BytesIO(reader.encode("utf-8"))
        self.reader = reader
        self.location = SourceLocation(1, 1)
        self.saved_ch = "\x00"
        self.saved_location = SourceLocation(1, 1)
        self.saved_token = None
        self.spaces = 0

    # ---- characters (src/scene.rs:150-183) ----
    def _update_pos(self, ch):
        if ch == "\n":
            self.location = SourceLocation(self.location.line_num + 1, 1)
        else:
            self.location = SourceLocation(self.location.line_num, self.location.col_num + 1)

    def read_char(self):
        if self.saved_ch != "\x00":
            ch, self.saved_ch = self.saved_ch, "\x00"
        else:
            b = self.reader.read(1)
            ch = chr(b[0]) if b else "\x00"        # eof reads as NUL; a Stop token is raised later
        self.saved_location = self.location
        self._update_pos(ch)
        return ch

    def unread_char(self, ch):
        self.saved_ch = ch
        self.location = self.saved_location

    def skip_comment(self):
        ch = self.read_char()
        while ch == " ":
            ch = self.read_char()
        self.unread_char(ch)
        if ch == "#":
            self.read_char()
            while True:
                ch = self.read_char()
                if ch in "\n\x00":
                    self.unread_char(ch)
                    break

    def skip_whitespaces_and_comments(self):
        ch = self.read_char()
        while ch in " \n#":
            if ch == "#":
                while True:
                    ch = self.read_char()
                    if ch in "\n\x00":
                        break
            ch = self.read_char()
        self.unread_char(ch)

    def count_spaces(self):
        self.spaces = 1
        ch = self.read_char()
        while ch == " ":
            self.spaces += 1
            ch = self.read_char()
        self.unread_char(ch)

    # ---- tokens (src/scene.rs:268-422) ----
    def _parse_string(self, loc, delimiter):
        out = []
        while True:
            ch = self.read_char()
            if ch == delimiter:
                break
            if ch in "\x00\n":
                self.unread_char(ch)
                raise SceneErr("UnclosedString", self.location, f"unclosed `{delimiter}` untermineted string")
            out.append(ch)
        return Token("string", loc, "".join(out))

    def _parse_float(self, first, loc):
        tok = [first]
        while True:
            ch = self.read_char()
            if ch in "eE":
                nx = self.read_char()
                if not (nx.isdigit() and nx.isascii() or nx in "+-"):
                    # only one character can be pushed back: the exponent letter is what the next
                    # token starts with (the reference's second unread overwrites the first)
                    self.unread_char(nx)
                    self.unread_char(ch)
                    break
                tok += [ch, nx]
            elif ch.isascii() and ch.isdigit() or ch == ".":
                tok.append(ch)
            else:
                self.unread_char(ch)
                break
        text = "".join(tok)
        try:
            return Token("number", loc, parse_f32(text))
        except ValueError:
            raise SceneErr("FloatParseFailure", loc, f"{text!r} is an invalid floating-point number")

    def _parse_keyword_or_identifier(self, first, loc):
        tok = [first]
        while True:
            ch = self.read_char()
            if not (ch.isascii() and ch.isalnum() or ch == "_"):
                self.unread_char(ch)
                break
            tok.append(ch)
        text = "".join(tok)
        return Token("keyword" if text in KEYWORDS else "identifier", loc, text)

    def read_token(self):
        if self.saved_token is not None:
            tok, self.saved_token = self.saved_token, None
            return tok
        loc = self.location
        ch = self.read_char()
        if ch == "\x00":
            return Token("stop", loc)
        if ch in SYMBOLS:
            nx = self.read_char()
            self.unread_char(nx)
            if ch == "-" and (nx.isascii() and nx.isdigit() or nx == "."):
                return self._parse_float(ch, loc)
            return Token("symbol", loc, ch)
        if ch.isascii() and ch.isdigit() or ch in "+.":
            return self._parse_float(ch, loc)
        if ch in "\"'":
            return self._parse_string(loc, ch)
        if ch.isascii() and ch.isalpha() or ch == "_":
            return self._parse_keyword_or_identifier(ch, loc)
        raise SceneErr("InvalidCharacter", loc, f"{ch} invalid character")

    def unread_token(self, token):
        self.saved_token = token

    # ---- matchers (src/scene.rs:428-588) ----
    @staticmethod
    def _not_match(token, expected):
        return SceneErr("NotMatch", token.loc, f"found {token.shown()!r} expected {expected}")

    def match_symbol(self, symbol):
        tok = self.read_token()
        if not (tok.kind == "symbol" and tok.value == symbol):
            raise self._not_match(tok, repr(symbol))

    def match_eol_or_inline_comment(self):
        tok = self.read_token()
        if tok.kind == "symbol" and tok.value == "\n":
            return
        if tok.kind == "symbol" and tok.value == " ":
            self.skip_comment()
            self.match_symbol("\n")
            return
        raise self._not_match(tok, "inline comment or '\\n'")

    def match_whitespaces_and_comments(self):
        tok = self.read_token()
        if tok.kind in ("keyword", "stop"):
            self.unread_token(tok)
        elif tok.kind == "symbol" and tok.value == "#":
            self.unread_char("#")
            self.skip_whitespaces_and_comments()
        else:
            self.skip_whitespaces_and_comments()

    def match_spaces(self, level, nested):
        for _ in range(self.spaces + level * 2 + self.spaces * nested):
            self.match_symbol(" ")

    def match_keyword(self, keyword):
        tok = self.read_token()
        if tok.kind == "keyword" and tok.value == keyword:
            self.match_symbol(":")
            return
        raise self._not_match(tok, keyword)

    def match_keywords(self, keywords):
        tok = self.read_token()
        if tok.kind == "keyword" and tok.value in keywords:
            self.match_symbol(":")
            return tok.value
        raise self._not_match(tok, "[" + ", ".join(keywords) + "]")

    def match_identifier(self):
        self.match_symbol(" ")
        tok = self.read_token()
        if tok.kind in ("identifier", "keyword"):     # a name may coincide with a keyword
            return tok.loc, tok.value
        raise self._not_match(tok, "identifier")

    def match_string(self):
        tok = self.read_token()
        if tok.kind == "string":
            return tok.loc, tok.value
        raise self._not_match(tok, "string")

    def match_number(self):
        tok = self.read_token()
        if tok.kind == "number":
            return tok.value
        raise self._not_match(tok, "floating-point number")

    def match_number_cli(self, cli):
        tok = self.read_token()
        if tok.kind == "number":
            return tok.value
        if tok.kind == "identifier":
            if tok.value == "RATIO":
                return F(cli.aspect_ratio)
            if tok.value == "DISTANCE":
                return F(1.0)
            raise SceneErr("UndefinedIdentifier", tok.loc,
                           f"{tok.value!r} floating-point number not defined, available [DISTANCE, RATIO]")
        raise self._not_match(tok, "floating-point number")

    # ---- values (src/scene.rs:593-657) ----
    def _triple(self):
        a = self.match_number()
        self.match_symbol(",")
        self.match_symbol(" ")
        b = self.match_number()
        self.match_symbol(",")
        self.match_symbol(" ")
        c = self.match_number()
        self.match_symbol("]")
        return (float(a), float(b), float(c))

    def parse_color(self, var):
        tok = self.read_token()
        if tok.kind == "symbol" and tok.value == "[":
            return self._triple()
        if tok.kind in ("identifier", "keyword"):
            if tok.value not in var.colors:
                raise SceneErr("UndefinedIdentifier", tok.loc, f"{tok.value!r} color not defined")
            return var.colors[tok.value]
        raise self._not_match(tok, "rgb color")

    def parse_vector(self, var):
        tok = self.read_token()
        if tok.kind == "symbol" and tok.value == "[":
            return self._triple()
        if tok.kind == "identifier":
            if tok.value not in var.vectors:
                raise SceneErr("UndefinedIdentifier", tok.loc, f"{tok.value!r} vector not defined, available [E1, E2, E3]")
            return var.vectors[tok.value]
        raise self._not_match(tok, "xyz vector")

    # ---- list blocks share one shape: first item, then items while the next line is indented ----
    def _block_items(self, parse_item):
        self.match_eol_or_inline_comment()
        self.match_symbol(" ")                      # at least one space of indent
        self.count_spaces()
        self.match_symbol("-")
        self.match_symbol(" ")
        parse_item()
        while True:
            self.match_eol_or_inline_comment()
            nx = self.read_token()
            self.unread_token(nx)
            if not (nx.kind == "symbol" and nx.value == " "):
                break
            self.match_spaces(0, 0)
            self.match_symbol("-")
            self.match_symbol(" ")
            parse_item()

    # ---- colors (src/scene.rs:662-718) ----
    def parse_colors(self, var):
        colors = {}

        def item():
            self.match_keyword("name")
            _, name = self.match_identifier()
            self.match_eol_or_inline_comment()
            self.match_spaces(1, 0)
            self.match_keyword("color")
            self.match_symbol(" ")
            colors[name] = self.parse_color(var)

        self._block_items(item)
        return colors

    # ---- materials (src/scene.rs:723-873) ----
    def parse_pigment(self, nested, var):
        self.match_spaces(1, nested)
        kind = self.match_keywords(["uniform", "checkered", "image"])
        self.match_symbol(" ")
        if kind == "uniform":
            return UniformPigment(self.parse_color(var))
        if kind == "image":
            loc, path = self.match_string()
            try:
                return ImagePigment(HdrImage.read_pfm_file(path))
            except (HdrImageErr, OSError) as e:
                raise SceneErr("PfmFileReadFailure", loc, f"{path!r} pfm file read failure: {e}")
        self.match_symbol("[")
        c1 = self.parse_color(var)
        self.match_symbol(",")
        self.match_symbol(" ")
        c2 = self.parse_color(var)
        self.match_symbol(",")
        self.match_symbol(" ")
        steps = _f32_to_u32_sat(self.match_number())
        self.match_symbol("]")
        return CheckeredPigment(c1, c2, steps)

    def parse_brdf(self, var):
        self.match_spaces(1, 0)
        kind = self.match_keywords(["diffuse", "specular"])
        self.match_eol_or_inline_comment()
        pigment = self.parse_pigment(1, var)
        return DiffuseBRDF(pigment) if kind == "diffuse" else SpecularBRDF(pigment, math.pi / 1800.0)

    def parse_materials(self, var):
        materials = {}

        def item():
            self.match_keyword("name")
            _, name = self.match_identifier()
            self.match_eol_or_inline_comment()
            brdf = self.parse_brdf(var)
            self.match_eol_or_inline_comment()
            materials[name] = Material(brdf, self.parse_pigment(0, var))

        self._block_items(item)
        return materials

    # ---- transformations (src/scene.rs:878-1050) ----
    def parse_transformation(self, transformations, var):
        tok = self.read_token()
        if tok.kind == "keyword":
            ch = self.read_char()
            self.unread_char(ch)
            if ch == ":":
                if tok.value not in ("rotationx", "rotationy", "rotationz", "scaling", "translation"):
                    raise self._not_match(tok, "[rotationx, rotationy, rotationz, scaling, translation]")
                self.match_symbol(":")
                self.match_symbol(" ")
                if tok.value == "rotationx":
                    return rotation_x(to_radians(self.match_number()))
                if tok.value == "rotationy":
                    return rotation_y(to_radians(self.match_number()))
                if tok.value == "rotationz":
                    return rotation_z(to_radians(self.match_number()))
                if tok.value == "scaling":
                    return scaling(self.parse_vector(var))
                return translation(self.parse_vector(var))
        if tok.kind in ("keyword", "identifier"):
            if tok.value not in transformations:
                raise SceneErr("UndefinedIdentifier", tok.loc, f"{tok.value!r} transformation not defined")
            return transformations[tok.value]
        raise self._not_match(tok, "transformation")

    def parse_composed_transformation(self, transformations, var):
        # only transformations defined earlier IN THIS BLOCK are visible to a compose list
        # (src/scene.rs:969 passes the block's own map; `var` only supplies the vectors E1..E3)
        visible = transformations
        self.match_keyword("name")
        _, name = self.match_identifier()
        self.match_eol_or_inline_comment()
        self.match_spaces(1, 0)
        self.match_keyword("compose")
        self.match_eol_or_inline_comment()
        self.match_spaces(1, 1)
        self.match_symbol("-")
        self.match_symbol(" ")
        # later entries multiply on the LEFT (src/scene.rs:969, :993)
        tr = self.parse_transformation(visible, var) * Transformation()
        while True:
            self.match_eol_or_inline_comment()
            nx = self.read_token()
            if not (nx.kind == "symbol" and nx.value == " "):
                self.unread_token(nx)
                break
            self.unread_token(nx)
            self.match_spaces(0, 0)
            nx2 = self.read_token()
            if nx2.kind == "symbol" and nx2.value == " ":
                self.unread_token(nx2)
                self.match_spaces(1, 0)
                self.match_symbol("-")
                self.match_symbol(" ")
                tr = self.parse_transformation(visible, var) * tr
            elif nx2.kind == "symbol" and nx2.value == "-":
                self.unread_token(nx2)              # a new named transformation starts here
                break
            else:
                raise self._not_match(nx2, "[' ', '-']")
        transformations[name] = tr

    def parse_transformations(self, var):
        transformations = {}
        self.match_eol_or_inline_comment()
        self.match_symbol(" ")
        self.count_spaces()
        self.match_symbol("-")
        self.match_symbol(" ")
        self.parse_composed_transformation(transformations, var)
        while True:
            nx = self.read_token()
            if nx.kind == "symbol" and nx.value == "-":
                self.match_symbol(" ")
                self.parse_composed_transformation(transformations, var)
            else:
                self.unread_token(nx)
                break
        return transformations

    # ---- shapes (src/scene.rs:1052-1140) ----
    def parse_shape(self, var):
        self.match_keyword("shape")
        self.match_symbol(" ")
        tok = self.read_token()
        if not (tok.kind == "keyword" and tok.value in ("plane", "sphere")):
            raise self._not_match(tok, "[plane, sphere]")
        self.match_eol_or_inline_comment()
        self.match_spaces(1, 0)
        self.match_keyword("material")
        loc, mid = self.match_identifier()
        if mid not in var.materials:
            raise SceneErr("UndefinedIdentifier", loc, f"{mid!r} material not defined")
        self.match_eol_or_inline_comment()
        self.match_spaces(1, 0)
        self.match_keyword("transformation")
        loc, tid = self.match_identifier()
        if tid not in var.transformations:
            raise SceneErr("UndefinedIdentifier", loc, f"{tid!r} transformation not defined")
        cls = Plane if tok.value == "plane" else Sphere
        return cls(var.transformations[tid], var.materials[mid])

    def parse_shapes(self, var):
        world = World()
        self._block_items(lambda: world.add(self.parse_shape(var)))
        return world

    # ---- camera (src/scene.rs:1142-1222) ----
    def parse_camera(self, var, cli):
        self.match_eol_or_inline_comment()
        self.match_symbol(" ")
        self.count_spaces()
        self.match_keyword("type")
        self.match_symbol(" ")
        loc, kind = self.match_string()
        if kind not in ("orthogonal", "perspective"):
            raise SceneErr("InvalidCamera", loc, f"found {kind!r} camera expected [\"orthogonal\", \"perspective\"]")
        self.match_eol_or_inline_comment()
        self.match_spaces(0, 0)
        self.match_keyword("ratio")
        self.match_symbol(" ")
        ratio = self.match_number_cli(cli)
        self.match_eol_or_inline_comment()
        distance = F(1.0)
        if kind == "perspective":
            self.match_spaces(0, 0)
            self.match_keyword("distance")
            self.match_symbol(" ")
            distance = self.match_number_cli(cli)
            self.match_eol_or_inline_comment()
        self.match_spaces(0, 0)
        self.match_keyword("transformation")
        loc, tid = self.match_identifier()
        if tid not in var.transformations:
            raise SceneErr("UndefinedIdentifier", loc, f"{tid!r} transformation not defined")
        tr = rotation_z(to_radians(cli.angle_deg)) * var.transformations[tid]      # src/scene.rs:1202
        if kind == "orthogonal":
            return OrthogonalCamera(float(ratio), tr)
        return PerspectiveCamera(float(distance), float(ratio), tr)

    # ---- scene (src/scene.rs:1240-1308) ----
    def parse_scene(self, cli):
        var, scene = Var(), Scene()
        blocks = ["camera", "colors", "materials", "shapes", "transformations"]
        while scene.camera is None or scene.shapes is None:
            self.match_whitespaces_and_comments()
            block = self.match_keywords(blocks)
            if block == "camera":
                scene.camera = self.parse_camera(var, cli)
            elif block == "colors":
                var.colors.update(self.parse_colors(var))
            elif block == "materials":
                var.materials.update(self.parse_materials(var))
            elif block == "shapes":
                scene.shapes = self.parse_shapes(var)
            else:
                var.transformations.update(self.parse_transformations(var))
            blocks.remove(block)                     # a block may appear once
        return scene
