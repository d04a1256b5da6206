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
    print(f"{name:24s} kernel {best:.3f} ms  rays {st.rays}")
