"""C4 / C5 frames with and without the hierarchy (diagnostics)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustracer_amd as ra
from rustracer_amd import _abi
from rustracer_amd.scenes import flake_camera, flake_world, random_spheres_camera, random_spheres_world
from rustracer_amd.world import flatten_camera
lib = _abi.lib(); _abi.check(lib.rt_init(0))
def run(name, world, cam, w, h, aa=1):
    dev = world.device()
    r = ra.PathTracer(world, ra.BLACK, 10, 3, 3).flatten(); c = flatten_camera(cam)
    out = torch.zeros((h, w, 3), device="cuda"); st = _abi.rt_stats()
    t0 = time.perf_counter()
    _abi.check(lib.rt_render(dev.handle, C.byref(c), C.byref(r), w, h, 42, 54, aa, None, out.data_ptr(), None, C.byref(st)))
    dt = time.perf_counter() - t0
    print(json.dumps({"config": name, "bvh": not os.environ.get("RUSTRACER_NO_BVH"), "w": w, "h": h, "ms": st.kernel_ms, "rays": st.rays, "rays_per_s": st.rays / (st.kernel_ms * 1e-3), "checksum": float(out.double().sum())}), flush=True)
which = sys.argv[1]
if which == "c4":
    fl = flake_world(True); run("C4 flake", fl, flake_camera(int(sys.argv[2]), int(sys.argv[3])), int(sys.argv[2]), int(sys.argv[3]))
else:
    rs = random_spheres_world(10000); run("C5 10002", rs, random_spheres_camera(int(sys.argv[2]), int(sys.argv[3])), int(sys.argv[2]), int(sys.argv[3]))
