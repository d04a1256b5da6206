"""Single-frame latency of the headline frame against the number of resident waves per SIMD and the pass-0 pixel order
(kernel ms from rt_stats, best of 5): does a frame that cannot fill the chip finish sooner on fewer, fuller waves?"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustracer_amd as ra
from rustracer_amd import _abi
from rustracer_amd.scenes import demo_camera, demo_world
from rustracer_amd.world import flatten_camera
sizes = [(1920, 1080), (1280, 720)]
world = demo_world(); rr = ra.PathTracer(world, ra.BLACK, 10, 3, 3).flatten()
dev = world.device(); lib = _abi.lib()
for W, H in sizes:
    cam = flatten_camera(demo_camera(W, H)); img = torch.zeros((H, W, 3), device="cuda")
    for sched_name, sched in (("one_pass", (0,)), ("short_passes", ())):
        dev.set_schedule(sched)
        for per_cu in (1, 2, 3, 4):
            for no_perm in (0, 1):
                os.environ["RUSTRACER_BLOCKS_PER_CU"] = str(per_cu)
                if no_perm: os.environ["RUSTRACER_NO_PERM"] = "1"
                else: os.environ.pop("RUSTRACER_NO_PERM", None)
                best = 1e9
                for _ in range(5):
                    st = _abi.rt_stats()
                    _abi.check(lib.rt_render(dev.handle, C.byref(cam), C.byref(rr), W, H, 42, 54, 1, None, img.data_ptr(), None, C.byref(st)))
                    best = min(best, st.kernel_ms)
                print(json.dumps({"size": f"{W}x{H}", "schedule": sched_name, "waves_per_simd": per_cu, "row_major": no_perm,
                                  "kernel_ms": round(best, 3), "rays": st.rays}), flush=True)
