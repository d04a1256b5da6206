"""Per-pass / per-wave timeline of one headline frame (diagnostics): RUSTRACER_DEBUG_WAVES dump -> summary."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustracer_amd as ra
from rustracer_amd import _abi
if os.environ.get("RT_LIB"): _abi.LIB_PATH = os.environ["RT_LIB"]
from rustracer_amd.scenes import demo_camera, demo_world, flake_camera, flake_world, random_spheres_camera, random_spheres_world
from rustracer_amd.world import flatten_camera

W, H = int(os.environ.get("RW", 1920)), int(os.environ.get("RH", 1080))
AA = int(os.environ.get("RAA", 1))
SCENE = os.environ.get("RSCENE", "demo")
if SCENE == "c4":
    world = flake_world(True); cam = flatten_camera(flake_camera(W, H))
elif SCENE == "c5":
    world = random_spheres_world(10000); cam = flatten_camera(random_spheres_camera(W, H))
else:
    world = demo_world(); cam = flatten_camera(demo_camera(W, H))
rr = ra.PathTracer(world, ra.BLACK, 10, 3, 3).flatten()
NF = int(os.environ.get("RFRAMES", 1))
dev = world.device(); img = torch.zeros((NF, H, W, 3), device="cuda")
cams = (_abi.rt_camera * NF)(*([cam] * NF)); outs = (C.c_void_p * NF)(*[img[b].data_ptr() for b in range(NF)])
lib = _abi.lib()
path = "/tmp/waves.bin"
os.environ["RUSTRACER_DEBUG_WAVES"] = path
os.environ["RUSTRACER_DEBUG_PASSES"] = "1"
if os.environ.get("RONEPASS"): dev.set_schedule((0,))
for rep in range(int(os.environ.get("RREPS", 3))):
    st = _abi.rt_stats()
    _abi.check(lib.rt_render_frames(dev.handle, cams, NF, C.byref(rr), W, H, 42, 54, AA, None, outs, None, C.byref(st)))
raw = np.fromfile(path, dtype=np.uint64)
n_pass, n_waves = int(raw[0]), int(raw[1])
d = raw[2:].reshape(n_pass, n_waves, 17).astype(np.float64)
print(f"kernel_ms={st.kernel_ms:.3f} rays={st.rays} passes={n_pass}")
T0 = d[0, :, 0][d[0, :, 0] > 0].min()
tot_rounds = 0
for p in range(n_pass):
    t0, t1, rounds, lanes, team = d[p].T[:5]
    sec = d[p][:, 5:]
    m = t1 > 0
    if not m.any():
        print(f"pass {p}: empty"); continue
    t0, t1, rounds, lanes, team = t0[m], t1[m], rounds[m], lanes[m], team[m]
    sec = sec[m].sum(0) / max(rounds.sum(), 1)
    life = (t1 - t0) / 100.0
    tot_rounds += rounds.sum()
    print(f"pass {p}: T={int(team.max())} waves {m.sum()} start {(t0.min()-T0)/100:.0f} us end {(t1.max()-T0)/100:.0f} us | "
          f"life mean {life.mean():.0f} p90 {np.percentile(life,90):.0f} max {life.max():.0f} us | rounds mean {rounds.mean():.0f} max {rounds.max():.0f} "
          f"| wave-rounds {rounds.sum():.3e} lanes/round {lanes.sum()/max(rounds.sum(),1):.1f} | us/round {(life.sum()/max(rounds.sum(),1)):.2f}")
    print("        cycles/round: generate %.0f scatter %.0f scan %.0f shade-rest %.0f fold %.0f park/rest %.0f | hit+pigment %.0f roulette %.0f leaf/skip %.0f open %.0f push %.0f pop %.0f  (total %.0f)" % (tuple(sec) + (sec.sum(),)))
print(f"total wave-rounds {tot_rounds:.3e}")
if os.environ.get("HELP_DEBUG"):
    o = np.argsort(-d[0, :, 2])[:12]
    for w in o:
        r = d[0, w]
        print(f"wave {w}: rounds {int(r[2])} lanes/round {r[3]/max(r[2],1):.1f} life {(r[1]-r[0])/100:.0f} us | exhausted rounds {int(r[13])} of which <=16 live {int(r[14])}, with a last-level diffuse lane {int(r[16])}, engaged {int(r[15])}")
