"""Can the render kernel store its pixels straight into memory across a link?  20 frames in one launch into (a) HBM,
(b) pinned host memory over PCIe (the only remote memory a one-GPU box has).  Kernel ms, best of 3."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustracer_amd as ra
from rustracer_amd import _abi, scenes
from rustracer_amd.world import flatten_camera
W, H, NF = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = _abi.lib(); _abi.check(lib.rt_init(0))
world = scenes.demo_world(); cam = flatten_camera(scenes.demo_camera(W, H))
rr = ra.PathTracer(world, ra.BLACK, 10, 3, 3).flatten()
dev = world.device(); dev.set_schedule((0,))
cams = (_abi.rt_camera * NF)(*([cam] * NF))
for name, img in (("hbm", torch.zeros((NF, H, W, 3), device="cuda")), ("pinned host over PCIe", torch.zeros((NF, H, W, 3), pin_memory=True))):
    outs = (C.c_void_p * NF)(*[img[b].data_ptr() for b in range(NF)])
    best = 1e9
    for _ in range(3):
        st = _abi.rt_stats()
        _abi.check(lib.rt_render_frames(dev.handle, cams, NF, C.byref(rr), W, H, 42, 54, 1, None, outs, None, C.byref(st)))
        best = min(best, st.kernel_ms)
    ok = bool(torch.isfinite(img).all())
    print(json.dumps({"target": name, "frames": NF, "kernel_ms": round(best, 3), "GB_per_s_stored": round(NF * W * H * 12 / best / 1e6, 1), "finite": ok}), flush=True)
