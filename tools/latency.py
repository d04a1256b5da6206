"""Single-frame latency of the headline frame under both schedules (kernel ms from rt_stats, best of 5)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustracer_amd as ra
from rustracer_amd import _abi
from rustracer_amd.scenes import demo_camera, demo_world
from rustracer_amd.world import flatten_camera
W, H = 1920, 1080
world = demo_world(); cam = flatten_camera(demo_camera(W, H)); rr = ra.PathTracer(world, ra.BLACK, 10, 3, 3).flatten()
dev = world.device(); img = torch.zeros((H, W, 3), device="cuda"); lib = _abi.lib()
for name, sched in (("short passes + teams", ()), ("one pass (FAST)", (0,))):
    dev.set_schedule(sched)
    best = 1e9
    for _ in range(6):
        st = _abi.rt_stats()
        _abi.check(lib.rt_render(dev.handle, C.byref(cam), C.byref(rr), W, H, 42, 54, 1, None, img.data_ptr(), None, C.byref(st)))
        best = min(best, st.kernel_ms)
    print(f"{name:24s} kernel {best:.3f} ms  rays {st.rays} shared {st.rays_shared}")
# the literal drop-in call: host buffers in, host image out (scene upload, render, 24.9 MB D2H)
import time
import numpy as np
from rustracer_amd.world import flatten_world
flat = flatten_world(world)
out = np.zeros((H, W, 3), dtype=np.float32)
best = 1e9
for _ in range(5):
    st = _abi.rt_stats()
    t0 = time.perf_counter()
    _abi.check(lib.rt_fire_all_rays(*flat.args(), C.byref(cam), C.byref(rr), W, H, 42, 54, 1, out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st)))
    best = min(best, (time.perf_counter() - t0) * 1e3)
print(f"rt_fire_all_rays (host buffers) wall {best:.3f} ms, of which kernel {st.kernel_ms:.3f} ms")
