"""Rays/s and path samples/s of the other BASELINE configs on one GPU (scene resident in HBM).
C1 onoff 640x480, C2 flat 1280x960 aa4, C4 sphere-flake (228 shapes, checkered + image pigments)
pathtracer 3840x2160, C5 10 002 shapes pathtracer at a reduced resolution (brute-force scan)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustracer_amd as ra
from rustracer_amd import _abi
from rustracer_amd.scenes import (demo_camera, demo_world, flake_camera, flake_world, random_spheres_camera,
                                  random_spheres_world)
from rustracer_amd.world import flatten_camera

lib = _abi.lib()
_abi.check(lib.rt_init(0))


def run(name, world, cam, renderer, w, h, aa, frames, reps, flops_per_ray):
    dev = world.device()
    dev.set_schedule((0,) if frames > 1 else ())
    c, r = flatten_camera(cam), renderer.flatten()
    cams = (_abi.rt_camera * frames)(*([c] * frames))
    outs = [torch.zeros((h, w, 3), dtype=torch.float32, device="cuda") for _ in range(frames)]
    ptrs = (C.c_void_p * frames)(*[o.data_ptr() for o in outs])
    st = _abi.rt_stats()
    s = torch.cuda.current_stream().cuda_stream
    _abi.check(lib.rt_render_frames(dev.handle, cams, frames, C.byref(r), w, h, 42, 54, aa, None, ptrs, s, C.byref(st)))
    t0 = time.perf_counter()
    for _ in range(reps):
        _abi.check(lib.rt_render_frames(dev.handle, cams, frames, C.byref(r), w, h, 42, 54, aa, None, ptrs, s, None))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out = {"config": name, "shapes": len(world.shapes), "width": w, "height": h, "aa": aa, "frames_per_launch": frames,
           "ms_per_frame": dt / frames * 1e3, "path_samples_per_s": w * h * aa * aa * frames / dt,
           "rays_per_frame": st.rays / frames, "rays_per_s": st.rays / dt,
           "algorithmic_tflops": st.rays * flops_per_ray / dt / 1e12}
    print(json.dumps(out), flush=True)


demo = demo_world()
run("C1 demo onoff 640x480 aa1", demo, demo_camera(640, 480), ra.OnOffRenderer(demo, ra.BLACK, ra.WHITE), 640, 480, 1, 8, 20, 220)
run("C2 demo flat 1280x960 aa4", demo, demo_camera(1280, 960), ra.FlatRenderer(demo, ra.BLACK), 1280, 960, 4, 2, 10, 220 + 62)
# C4 / C5: the linear scan of World::ray_intersection (the roofline figure: every ray tests every shape) and the
# exact-winner hierarchy (same image bit for bit; its skipped tests are NOT credited as algorithmic flops)
for linear in (True, False):
    if linear:
        os.environ["RUSTRACER_NO_BVH"] = "1"
    else:
        os.environ.pop("RUSTRACER_NO_BVH", None)
    how = "linear scan" if linear else "hierarchy"
    fl = flake_world(True)
    fl.device()                       # the scene (and its hierarchy, if enabled) is built here
    run(f"C4 flake (228 shapes, checkered + image pigments) pathtracer n=10 d=3 3840x2160 aa1, {how}", fl,
        flake_camera(3840, 2160), ra.PathTracer(fl, ra.BLACK, 10, 3, 3), 3840, 2160, 1, 1, 2,
        (60 * 227 + 40 + 62 + 46) if linear else 0)
    rs = random_spheres_world(10000)
    rs.device()
    run(f"C5 10002 shapes pathtracer n=10 d=3 960x540 aa1 (1/64 of 7680x4320, 1/16 of the samples per pixel), {how}", rs,
        random_spheres_camera(960, 540), ra.PathTracer(rs, ra.BLACK, 10, 3, 3), 960, 540, 1, 1, 1,
        (60 * 10001 + 40 + 62 + 46) if linear else 0)
    if not linear:
        run(f"C5 10002 shapes pathtracer n=10 d=3 1920x1080 aa1, {how}", rs, random_spheres_camera(1920, 1080),
            ra.PathTracer(rs, ra.BLACK, 10, 3, 3), 1920, 1080, 1, 1, 1, 0)
