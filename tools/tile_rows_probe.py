"""Cost of a row-tile partition on the hierarchy scenes: kernel ms of the whole frame vs the sum and max of its parts for
several tile heights (one GPU renders the parts one after the other).  python tools/tile_rows_probe.py c5|c4 W H AA N_PARTS"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustracer_amd as ra
from rustracer_amd import _abi, scenes
from rustracer_amd.world import flatten_camera

cfg, W, H, AA, NP = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
lib = _abi.lib(); _abi.check(lib.rt_init(0))
if cfg == "c5":
    world, cam = scenes.random_spheres_world(10000), flatten_camera(scenes.random_spheres_camera(W, H))
elif cfg == "c4":
    world, cam = scenes.flake_world(True), flatten_camera(scenes.flake_camera(W, H))
else:
    world, cam = scenes.demo_world(), flatten_camera(scenes.demo_camera(W, H))
rr = ra.PathTracer(world, ra.BLACK, 10, 3, 3).flatten()
dev = world.device(); dev.set_schedule((0,))
buf = torch.zeros((H * W, 3), dtype=torch.float32, device="cuda")


def render(part):
    best = None
    for _ in range(2):
        st = _abi.rt_stats()
        _abi.check(lib.rt_render(dev.handle, C.byref(cam), C.byref(rr), W, H, 42, 54, AA, C.byref(part) if part else None,
                                 buf.data_ptr(), None, C.byref(st)))
        best = st.kernel_ms if best is None else min(best, st.kernel_ms)
    return best, st.rays


whole, rays = render(None)
print(json.dumps({"config": cfg, "size": [W, H, AA], "whole_ms": whole, "rays": rays}), flush=True)
for tile in [int(t) for t in os.environ.get("TILES", "1,2,4,8,16,64").split(",")] + [(H + NP - 1) // NP]:
    ms, rs = zip(*[render(_abi.rt_partition(p, NP, tile, 0)) for p in range(NP)])
    print(json.dumps({"tile_rows": tile, "n_parts": NP, "sum_ms": sum(ms), "max_ms": max(ms), "sum_over_whole": sum(ms) / whole,
                      "speedup_if_parallel": whole / max(ms), "max_over_mean_rays": max(rs) / (sum(rs) / NP)}), flush=True)
