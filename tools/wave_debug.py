"""Per-wave timeline of one headline frame (diagnostics): RUSTRACER_DEBUG_WAVES dump -> summary."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustracer_amd as ra
from rustracer_amd import _abi
from rustracer_amd.scenes import demo_camera, demo_world
from rustracer_amd.world import flatten_camera

W, H = int(os.environ.get("RW", 1920)), int(os.environ.get("RH", 1080))
world = demo_world(); cam = flatten_camera(demo_camera(W, H)); rr = ra.PathTracer(world, ra.BLACK, 10, 3, 3).flatten()
dev = world.device(); img = torch.zeros((H, W, 3), device="cuda")
lib = _abi.lib()
for rep in range(3):
    path = "/tmp/waves.bin"
    os.environ["RUSTRACER_DEBUG_WAVES"] = path
    st = _abi.rt_stats()
    _abi.check(lib.rt_render(dev.handle, C.byref(cam), C.byref(rr), W, H, 42, 54, 1, None, img.data_ptr(), None, C.byref(st)))
d = np.fromfile(path, dtype=np.uint64).reshape(-1, 5).astype(np.float64)
t0, t1, rounds, lanes, refills = d.T
T0 = t0.min()
life = (t1 - t0) / 100.0  # us
print(f"T={os.environ.get('RUSTRACER_TEAM','default')} sort={'off' if os.environ.get('RUSTRACER_NO_SORT') else 'on'} kernel_ms={st.kernel_ms:.3f} rays={st.rays}")
print(f"waves {len(d)}  span {(t1.max()-T0)/100:.0f} us  start spread {(t0.max()-T0)/100:.0f} us")
print(f"lifetime us: mean {life.mean():.0f} p50 {np.percentile(life,50):.0f} p90 {np.percentile(life,90):.0f} max {life.max():.0f}")
print(f"rounds/wave: mean {rounds.mean():.0f} p50 {np.percentile(rounds,50):.0f} p90 {np.percentile(rounds,90):.0f} max {rounds.max():.0f}; total wave-rounds {rounds.sum():.3e}")
print(f"active lanes per round: {lanes.sum()/rounds.sum():.1f} of 64; max rounds of one pixel {refills.max():.0f} (p99 over waves {np.percentile(refills,99):.0f})")
print(f"us per round: mean {(life/np.maximum(rounds,1)).mean():.2f}; for longest-lived 1%: {(life/np.maximum(rounds,1))[life>=np.percentile(life,99)].mean():.2f}")
end = (t1 - T0) / 100.0
hist, edges = np.histogram(end, bins=10)
print("wave end-time histogram (us):", [f"{int(e)}:{c}" for e, c in zip(edges[1:], hist)])
