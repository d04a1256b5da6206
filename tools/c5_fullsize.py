"""BASELINE config 5 at its full size on one GPU: 10 002 shapes, 7680x4320, 16 samples per pixel, pathtracer n=10 d=3 rr=3,
through the exact-winner hierarchy.  The oracle cannot render this (a literal linear scan: months), so the run is checked
by size-independent properties: the whole frame in one launch (one-pass schedule) against the same frame rendered as 8
row-tile parts under the short-pass schedule and assembled by rt_unpack_parts -- images bit-equal, ray / sample / scatter
counters equal.  Appends JSON lines to gpurun_out/c5_fullsize.jsonl (copied to profiles/ by hand)."""
import ctypes as C, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustracer_amd as ra
from rustracer_amd import _abi
from rustracer_amd.scenes import random_spheres_camera, random_spheres_world
from rustracer_amd.world import flatten_camera

W, H, AA = int(os.environ.get("RW", 7680)), int(os.environ.get("RH", 4320)), int(os.environ.get("RAA", 4))
N_PARTS = 8
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "c5_fullsize.jsonl")
stop = False


def heartbeat():
    t0 = time.time()
    while not stop:
        time.sleep(30)
        print(f"[c5_fullsize] {time.time() - t0:.0f} s", flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
lib = _abi.lib()
_abi.check(lib.rt_init(0))
world = random_spheres_world(10000)
cam = flatten_camera(random_spheres_camera(W, H))
rr = ra.PathTracer(world, ra.BLACK, 10, 3, 3).flatten()
dev = world.device()
KEYS = ("rays", "samples", "diffuse_scatters", "specular_scatters")


def emit(d):
    print(json.dumps(d), flush=True)
    with open(OUT, "a") as f:
        f.write(json.dumps(d) + "\n")


whole = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
dev.set_schedule((0,))
st = _abi.rt_stats()
t0 = time.time()
_abi.check(lib.rt_render(dev.handle, C.byref(cam), C.byref(rr), W, H, 42, 54, AA, None, whole.data_ptr(), None, C.byref(st)))
whole_stats = {k: getattr(st, k) for k in KEYS}
whole_ms = st.kernel_ms
emit({"what": "C5 whole frame, one launch, one-pass schedule, hierarchy", "size": [W, H, AA], "kernel_ms": st.kernel_ms,
      "wall_s": time.time() - t0, **whole_stats, "rays_per_s": st.rays / (st.kernel_ms * 1e-3),
      "samples_per_s": st.samples / (st.kernel_ms * 1e-3)})

# the parts under both schedules (attribute the parts-vs-whole overhead to the schedule or to the split)
sizes = [lib.rt_partition_pixels(W, H, C.byref(_abi.rt_partition(p, N_PARTS, 1, 0))) for p in range(N_PARTS)]
stride = max(sizes)
all_equal = True
for sched_name, sched in (("default short-pass schedule", ()), ("one-pass schedule", (0,))):
    dev.set_schedule(sched)
    parts = torch.zeros((N_PARTS, stride, 3), dtype=torch.float32, device="cuda")
    tot = {k: 0 for k in KEYS}
    per_part_ms = []
    for p in range(N_PARTS):
        part = _abi.rt_partition(p, N_PARTS, 1, 0)
        st = _abi.rt_stats()
        _abi.check(lib.rt_render(dev.handle, C.byref(cam), C.byref(rr), W, H, 42, 54, AA, C.byref(part), parts[p].data_ptr(), None, C.byref(st)))
        for k in KEYS:
            tot[k] += getattr(st, k)
        per_part_ms.append(st.kernel_ms)
        print(f"[c5_fullsize] {sched_name}, part {p}: {st.kernel_ms:.0f} ms, {st.rays} rays", flush=True)
    image = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    _abi.check(lib.rt_unpack_parts(parts.data_ptr(), stride, N_PARTS, 1, W, H, image.data_ptr(), None))
    torch.cuda.synchronize()
    equal = bool(torch.equal(image.view(torch.int32), whole.view(torch.int32)))
    all_equal = all_equal and equal and tot == whole_stats
    emit({"what": f"C5 as {N_PARTS} row-tile parts (1-row tiles), {sched_name}, rt_unpack_parts", "size": [W, H, AA],
          "per_part_kernel_ms": per_part_ms, "max_over_mean_part_ms": max(per_part_ms) / (sum(per_part_ms) / N_PARTS),
          "sum_of_parts_over_whole": sum(per_part_ms) / whole_ms, "strong_scaling_efficiency_predicted": whole_ms / (N_PARTS * max(per_part_ms)), **tot,
          "image_bit_equal_to_whole_frame": equal, "counters_equal_to_whole_frame": tot == whole_stats,
          "finite": bool(torch.isfinite(whole).all()), "mean_radiance": float(whole.mean())})
    del parts, image
dev.set_schedule(())
stop = True
if not all_equal:
    raise SystemExit("C5 full size: the partitioned render differs from the whole-frame render")
