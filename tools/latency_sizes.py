"""Single-frame kernel latency, both schedules, several frame sizes (demo scene, pathtracer n=10 d=3)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustracer_amd as ra
from rustracer_amd import _abi
from rustracer_amd.scenes import demo_camera, demo_world
from rustracer_amd.world import flatten_camera
world = demo_world(); rr = ra.PathTracer(world, ra.BLACK, 10, 3, 3).flatten(); dev = world.device(); lib = _abi.lib()
for W, H in ((320, 180), (640, 360), (960, 540), (1280, 720), (1920, 1080), (3840, 2160)):
    cam = flatten_camera(demo_camera(W, H)); img = torch.zeros((H, W, 3), device="cuda")
    res = []
    for sched in ((), (0,)):
        dev.set_schedule(sched)
        best = 1e9
        for _ in range(5):
            st = _abi.rt_stats()
            _abi.check(lib.rt_render(dev.handle, C.byref(cam), C.byref(rr), W, H, 42, 54, 1, None, img.data_ptr(), None, C.byref(st)))
            best = min(best, st.kernel_ms)
        res.append(best)
    print(f"{W}x{H}: short passes + teams {res[0]:.3f} ms   one pass {res[1]:.3f} ms")
