"""What one rank of bench.py's weak-scaling run does, timed on one GPU: for N in 1, 2, 4, 8 the rank renders its rotating
row blocks of steps*N frames in the launches bench.py would issue.  mode direct (bench.py's default): pixels stored in
place into full-size images (here in this GPU's HBM; on a node they are rank 0's, across xGMI).  mode copy (--gather p2p):
blocks rendered into local buffers on 2 streams and copied device-to-device on a copy stream.  The xGMI hop itself is not
in these numbers (one GPU here); the render-side cost of the partition is.
python tools/weak_rank_probe.py [steps] [direct|copy]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustracer_amd as ra
from rustracer_amd import _abi, scenes
from rustracer_amd.world import flatten_camera

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
MODE = sys.argv[2] if len(sys.argv) > 2 else "direct"
W, H = 1920, 1080
lib = _abi.lib(); _abi.check(lib.rt_init(0))
world = scenes.demo_world(); cam = flatten_camera(scenes.demo_camera(W, H))
rr = ra.PathTracer(world, ra.BLACK, 10, 3, 3).flatten()
dev = world.device(); dev.set_schedule((0,))
base = None
for N in (1, 2, 4, 8):
    for rank in sorted({0, N - 1} if N > 1 else {0}):
        total = STEPS * N
        copy = MODE == "copy" and N > 1
        F = 2 if copy else 1
        cap = 128 * N
        n_l = max((total + cap - 1) // cap, min(4, STEPS) if copy else 1)
        B = min((total + n_l - 1) // n_l, cap)
        B = max(N, (B + N - 1) // N * N)
        block_rows = H // N
        block_px = block_rows * W
        streams = [torch.cuda.Stream() for _ in range(F)]
        cstream = torch.cuda.Stream()
        cams = (_abi.rt_camera * B)(*([cam] * B))
        if copy:
            local = [torch.zeros((B, block_px, 3), device="cuda") for _ in range(F + 1)]
            sink = torch.zeros((min(total, 256), block_px, 3), device="cuda")      # stand-in for rank 0's image blocks
        else:
            ring = min(total, 1024) // B * B or B
            images = torch.zeros((ring, H, W, 3), device="cuda")

        def run():
            f0, l, done = 0, 0, {}
            while f0 < total:
                n = min(B, total - f0)
                st = streams[l % F]
                part = _abi.rt_partition((rank + f0) % N, N, block_rows, 1, 0 if copy else 1)
                with torch.cuda.stream(st):
                    if copy:
                        if (l - (F + 1)) in done:
                            st.wait_event(done.pop(l - (F + 1)))
                        ptrs = (C.c_void_p * n)(*[local[l % (F + 1)][b].data_ptr() for b in range(n)])
                    else:
                        ptrs = (C.c_void_p * n)(*[images[(f0 + b) % ring].data_ptr() for b in range(n)])
                    _abi.check(lib.rt_render_frames(dev.handle, cams, n, C.byref(rr), W, H, 42, 54, 1, C.byref(part), ptrs, st.cuda_stream, None))
                    ev = torch.cuda.Event(); ev.record(st)
                if copy:
                    with torch.cuda.stream(cstream):
                        cstream.wait_event(ev)
                        for b in range(n):
                            sink[(f0 + b) % sink.shape[0]].copy_(local[l % (F + 1)][b], non_blocking=True)
                        done[l] = torch.cuda.Event(); done[l].record(cstream)
                f0 += n; l += 1
            return l

        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); nl = run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort(); med = ts[2]
        if base is None:
            base = med
        print(json.dumps({"mode": "copy" if copy else "direct", "n_gpus": N, "rank": rank, "steps": STEPS, "frames": total, "launches": nl,
                          "frames_per_launch": B, "streams": F, "rank_ms_median": round(med, 3), "render_side_efficiency": round(base / med, 3)}), flush=True)
        if copy:
            del local, sink
        else:
            del images
        torch.cuda.empty_cache()
