#!/usr/bin/env python3
"""bench.py — headline benchmark: path samples/s of examples/demo.yml, pathtracer, 1920x1080.

A step = one ImageTracer::fire_all_rays pass (one whole frame) with the scene already resident in
HBM and the image left in HBM.  Steps are issued as launches of B frames (rt_render_frames: the
frames of a launch share one pixel queue on the GPU; default: the steps dealt evenly over the fewest launches
of at most 128 frames) over F HIP streams (default 1; 2 with several GPUs, for the gather),
so B*F frames are in flight; every frame is rendered in full, nothing is cached between frames.  N GPUs split EVERY
frame by row tiles dealt round-robin (strong scaling), one process per GPU; every rank pushes its part of a
launch into rank 0's buffers (mapped through HIP IPC handles) with a device-to-device copy that runs beside the next
launch's render, a CPU-side barrier per launch tells rank 0 when the parts are complete, and rt_unpack_parts
assembles the images there (--gather rccl, also the fallback when the handles cannot be exchanged: one RCCL gather
per launch instead).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, AA = 1920, 1080, 1
NUM_OF_RAYS, MAX_DEPTH, RR_LIMIT = 10, 3, 3
INIT_STATE, INIT_SEQ = 42, 54
EXPECTED_RAYS = 44180520            # oracle (strict math) ray count of this exact frame; tests/test_gpu_parity.py
PEAK_FP32_VECTOR_TFLOPS = 157.3     # MI355X_MICROARCH.md: spec, packed FMA counted as 2 flop
PEAK_STRICT_NOFMA_TFLOPS = 78.6     # same issue rate with unfused mul/add (the reference forbids FMA contraction)
# HBM bytes per frame of the default configuration, from profiles/r01c_rocprof_default.txt (rocprofv3
# --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, mean KB per dispatch over 4 dispatches that render
# 1 + 24 + 2*120 = 265 frames: 191 and 1726500; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950):
# (2*191 + 1726500) KB * 4 / 265 = 26.1 MB.  The algorithmic minimum is the 24.9 MB image (12-byte pixel stores
# fill cache lines partially; the scene is read once per block into LDS); the path is VALU-bound, not HBM-bound.
TRAFFIC_BYTES_PER_FRAME = (2 * 191 + 1726500) * 1000 * 4 / 265


def partition_rows(height, part, n_parts, tile_rows):
    """Rows of `part` (host mirror of part_rows() in csrc/rt_host.inc)."""
    rows = []
    n_tiles = (height + tile_rows - 1) // tile_rows
    for j in range(part, n_tiles, n_parts):
        rows.extend(range(j * tile_rows, min((j + 1) * tile_rows, height)))
    return rows


def flops_per_ray(n_spheres, n_planes):
    """60 per sphere test, 40 per plane test, +62 hit tail, +46 shade."""
    return 60 * n_spheres + 40 * n_planes + 62 + 46


def cpu_baseline(flat, cam, rr, threads):
    """The CPU checker (oracle/, the reference's algorithm restated) on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        _, st = O.fire_all_rays(flat, cam, rr, WIDTH, HEIGHT, INIT_STATE, INIT_SEQ, AA, math_mode=O.MATH_LIBM,
                                n_threads=threads)
    dt = (time.perf_counter() - t0) / reps
    return {"value": WIDTH * HEIGHT * AA * AA / dt, "unit": "path samples/s", "cores": threads, "kind": "port",
            "sample": f"the full {WIDTH}x{HEIGHT} frame {reps} times ({st['rays']} rays each, {dt:.2f} s wall per frame "
                      "incl. serial all_rays)",
            "rays_per_s": st["rays"] / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--tile-rows", type=int, default=1)
    ap.add_argument("--inflight", type=int, default=0,
                    help="launches in flight on separate HIP streams (1..4; default 1, with several GPUs 2)")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="frames rendered by one rt_render_frames launch (default: the steps dealt evenly over the "
                         "fewest launches of at most 128 frames)")
    ap.add_argument("--schedule", choices=["auto", "passes", "onepass"], default="auto",
                    help="rt_scene_set_schedule: 'passes' = short passes (lowest single-frame latency), 'onepass' = one "
                         "pass per launch (best throughput with many frames in flight); auto = onepass unless "
                         "--frames-per-launch 1 on one stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: rehearse the N > 1 path on a box with fewer GPUs than ranks (parts staged through host memory)")
    ap.add_argument("--gather", choices=["rccl", "p2p"], default="p2p",
                    help="N > 1: how the parts reach rank 0.  rccl: one gather collective per launch.  p2p: rank 0 shares "
                         "its buffers through HIP IPC handles and every rank pushes its part with a device-to-device copy "
                         "(copy engine: runs beside the next launch's render, which an RCCL kernel cannot), with a "
                         "CPU-side barrier per launch; falls back to rccl if the handles cannot be exchanged")
    ap.add_argument("--verify", action="store_true",
                    help="after the run, rank 0 compares every assembled image with a whole-frame render, bit for bit")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import rustracer_amd as ra
    from rustracer_amd import _abi
    from rustracer_amd.scenes import demo_camera, demo_world
    from rustracer_amd.world import flatten_camera

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run")
    device = local_rank % torch.cuda.device_count() if args.dist_backend == "gloo" else local_rank
    torch.cuda.set_device(device)
    lib = _abi.lib()
    _abi.check(lib.rt_init(device))
    if world_size > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    scene = demo_world()
    flat = scene.flatten()
    cam = flatten_camera(demo_camera(WIDTH, HEIGHT))
    rr = ra.PathTracer(scene, ra.BLACK, NUM_OF_RAYS, MAX_DEPTH, RR_LIMIT).flatten()
    dev = scene.device()          # World resident in HBM before the timed region
    # one stream: a launch of this size fills the GPU by itself, and launches that share it run slower than one
    # after the other (measured: 240 steps 0.78 vs 0.79 ms/frame, 100 steps 0.79 vs 0.84); with several GPUs a
    # second stream lets the gather of one launch run beside the next launch
    F = max(1, min(args.inflight, 4)) if args.inflight else (2 if world_size > 1 else 1)
    if args.frames_per_launch:
        B = max(1, min(args.frames_per_launch, 512))
    else:
        # the fewest launches of at most 128 frames, the steps dealt evenly over them: every launch ends with one
        # drain of long pixels (measured, 240 steps: launches of 60 / 120 frames 0.78 / 0.75 ms/frame).  With
        # several GPUs at least 4 launches, so that the gather of one can overlap the render of the next.
        n_l = (args.steps + 127) // 128
        if world_size > 1:
            n_l = max(n_l, min(4, args.steps))
        B = max(1, min((args.steps + n_l - 1) // n_l, 128))
    onepass = args.schedule == "onepass" or (args.schedule == "auto" and (F > 1 or B > 1))
    dev.set_schedule((0,) if onepass else ())

    part = _abi.rt_partition(rank, world_size, args.tile_rows)
    sizes = [lib.rt_partition_pixels(WIDTH, HEIGHT, C.byref(_abi.rt_partition(p, world_size, args.tile_rows)))
             for p in range(world_size)]
    stride = max(sizes)
    # Launch l renders frames [l*B, l*B + n) on stream l % F with that slot's own buffers, so the long
    # single-pixel dependency chains at the end of one launch overlap the bulk of the next ones (and,
    # for N > 1, gather(l) overlaps render(l+1)).
    streams = [torch.cuda.Stream() for _ in range(F)]
    cams = (_abi.rt_camera * B)(*([cam] * B))
    # p2p gather: two more sets of gathered / image buffers than launches in flight, so that a part pushed by launch l
    # lands in a buffer whose previous reader (rank 0's unpack of launch l - NB) is known to have finished
    cpu_group, remote = None, None
    NB = F
    bufs = gathered = None
    if world_size > 1:
        bufs = [torch.zeros((B, stride, 3), dtype=torch.float32, device="cuda") for _ in range(F)]
        if args.gather == "p2p":
            from torch.multiprocessing.reductions import reduce_tensor
            NB = F + 2
            cpu_group = dist.new_group(backend="gloo") if args.dist_backend == "nccl" else dist.group.WORLD
            ok, box = 1, [None]
            if rank == 0:
                try:
                    gathered = [torch.zeros((world_size, B, stride, 3), dtype=torch.float32, device="cuda") for _ in range(NB)]
                    box[0] = [reduce_tensor(t) for t in gathered]
                except Exception as e:   # noqa: BLE001 - any failure means: use the collective
                    print(f"p2p gather unavailable on rank 0: {e}", file=sys.stderr)
                    box[0], ok = None, 0
            dist.broadcast_object_list(box, src=0, group=cpu_group)
            if rank == 0:
                remote = gathered
            elif box[0] is None:
                ok = 0
            else:
                try:
                    remote = [fn(*a) for fn, a in box[0]]     # rank 0's buffers mapped into this process
                    remote[0][rank, 0, :1].copy_(bufs[0][0, :1])  # a first write through the mapping, checked here
                    torch.cuda.synchronize()
                except Exception as e:   # noqa: BLE001
                    print(f"p2p gather unavailable on rank {rank}: {e}", file=sys.stderr)
                    ok = 0
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=cpu_group)
            if int(flag[0]) == 0:
                cpu_group, remote, gathered, NB = None, None, None, F
        if remote is None and rank == 0:
            gathered = [torch.zeros((world_size, B, stride, 3), dtype=torch.float32, device="cuda") for _ in range(F)]
    images = [torch.zeros((B, HEIGHT, WIDTH, 3), dtype=torch.float32, device="cuda") for _ in range(NB)]
    copy_done, unpack_done = {}, {}

    def out_ptrs(slot, n):
        t = bufs[slot] if world_size > 1 else images[slot]
        return (C.c_void_p * n)(*[t[b].data_ptr() for b in range(n)])

    def launch(l, n, ev=None, stats=None):
        slot = l % F
        st = streams[slot]
        with torch.cuda.stream(st):
            if ev:
                ev[0].record(st)
            _abi.check(lib.rt_render_frames(dev.handle, cams, n, C.byref(rr), WIDTH, HEIGHT, INIT_STATE, INIT_SEQ, AA,
                                            C.byref(part), out_ptrs(slot, n), st.cuda_stream, stats))
            if ev:
                ev[1].record(st)
            if remote is not None:
                # push this rank's part into rank 0's buffer (device-to-device copy on this stream); complete() does
                # the rest once every rank's copy has landed
                remote[l % NB][rank, :n].copy_(bufs[slot][:n], non_blocking=True)
                copy_done[l] = torch.cuda.Event()
                copy_done[l].record(st)
            elif world_size > 1:
                # the collective runs on RCCL's stream after st's render; work.wait() makes st (not the
                # host) wait for it, so the slot's buffers are free again when st reaches its next launch
                if args.dist_backend == "gloo":
                    host = bufs[slot].cpu()
                    hlist = [torch.empty_like(host) for _ in range(world_size)] if rank == 0 else None
                    dist.gather(host, gather_list=hlist, dst=0)
                    if rank == 0:
                        gathered[slot].copy_(torch.stack(hlist))
                elif rank == 0:
                    glist = list(gathered[slot].unbind(0))
                    dist.gather(bufs[slot], gather_list=glist, dst=0, async_op=True).wait()
                else:
                    dist.gather(bufs[slot], dst=0, async_op=True).wait()
                if rank == 0:
                    for b in range(n):
                        _abi.check(lib.rt_unpack_parts(gathered[slot][0, b].data_ptr(), B * stride, world_size,
                                                       args.tile_rows, WIDTH, HEIGHT, images[slot][b].data_ptr(),
                                                       st.cuda_stream))

    def complete(l, n):
        """p2p gather: launch l's parts are in rank 0's buffer l % NB once every rank's copy has finished (CPU barrier);
        rank 0 then assembles the images.  Rank 0 enters the barrier of launch l only after its unpack of launch l - 2
        has finished (its kernels run in the gaps the render leaves), which is what lets launch l + F overwrite that
        buffer."""
        copy_done.pop(l).synchronize()
        if rank == 0 and (l - 2) in unpack_done:
            unpack_done.pop(l - 2).synchronize()
        dist.barrier(group=cpu_group)
        if rank == 0:
            st = streams[l % F]
            for b in range(n):
                _abi.check(lib.rt_unpack_parts(gathered[l % NB][0, b].data_ptr(), B * stride, world_size, args.tile_rows,
                                               WIDTH, HEIGHT, images[l % NB][b].data_ptr(), st.cuda_stream))
            unpack_done[l] = torch.cuda.Event()
            unpack_done[l].record(st)

    def run(steps, events=None):
        sizes = []
        done = 0
        while done < steps:
            sizes.append(min(B, steps - done))
            done += sizes[-1]
        for l, n in enumerate(sizes):
            if remote is not None and l >= F:
                complete(l - F, sizes[l - F])
            launch(l, n, events[l] if events is not None else None)
        if remote is not None:
            for l in range(max(0, len(sizes) - F), len(sizes)):
                complete(l, sizes[l])
            unpack_done.clear()
        return len(sizes)

    def sync_all():
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # one instrumented single-frame launch: ray counters of this rank's part (deterministic, same every step)
    st = _abi.rt_stats()
    launch(0, 1, stats=C.byref(st))
    if remote is not None:
        complete(0, 1)
        unpack_done.clear()
    my_rays = st.rays
    if world_size == 1 and my_rays != EXPECTED_RAYS and not os.environ.get("RUSTRACER_BENCH_ABLATION"):
        raise SystemExit(f"ray counter {my_rays} != {EXPECTED_RAYS} (oracle, World::ray_intersection calls of this frame)")
    sync_all()

    run(args.warmup)
    sync_all()
    n_launches = (args.steps + B - 1) // B
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_launches)]
    t0 = time.perf_counter()
    run(args.steps, events)
    sync_all()
    dt = time.perf_counter() - t0

    # per-launch latency (start -> end of the launch's kernels on its own stream; launches in flight
    # stretch each other) and the device-side span of the whole timed region / launches = the effective
    # duration of one launch
    latency_ms = sum(a.elapsed_time(b) for a, b in events) / n_launches
    span_ms = max(events[0][0].elapsed_time(e[1]) for e in events[-F:])
    launch_ms = span_ms / n_launches
    t = torch.tensor([dt, launch_ms, float(my_rays), latency_ms], dtype=torch.float64, device="cuda")
    if world_size > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, total_rays = float(tmax[0]), float(tsum[2])
    else:
        total_rays = float(my_rays)

    if args.verify and rank == 0:
        ref = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.float32, device="cuda")
        _abi.check(lib.rt_render(dev.handle, C.byref(cam), C.byref(rr), WIDTH, HEIGHT, INIT_STATE, INIT_SEQ, AA, None,
                                 ref.data_ptr(), torch.cuda.current_stream().cuda_stream, None))
        torch.cuda.synchronize()
        n_last = args.steps - (n_launches - 1) * B     # frames of the last launch
        checked = 0
        for l in range(max(0, n_launches - NB), n_launches):   # the launches whose images are still in a buffer
            n_chk = n_last if l == n_launches - 1 else B
            for b in range(n_chk):
                if not torch.equal(images[l % NB][b].view(torch.int32), ref.view(torch.int32)):
                    raise SystemExit(f"verify: image {b} of launch {l} differs from the whole-frame render")
            checked += n_chk
        print(f"verify ok: {checked} images of the last {min(NB, n_launches)} launches bit-identical to a whole-frame render"
              + (" (p2p gather)" if remote is not None else ""), file=sys.stderr)

    if rank == 0:
        samples_per_step = WIDTH * HEIGHT * AA * AA
        value = samples_per_step * args.steps / dt
        fpr = flops_per_ray(dev_n_spheres(flat), dev_n_planes(flat))
        # roofline of the dominant kernel (k_pass) on THIS rank: algorithmic flops of one launch / its
        # effective duration (HIP events on the launch streams over the timed region)
        frames_per_launch = args.steps / n_launches
        achieved = my_rays * frames_per_launch * fpr / (launch_ms * 1e-3) / 1e12
        out = {
            "metric": "path samples/sec (node) demo.yml pathtracer 1080p",
            "value": value, "unit": "path samples/s", "n_gpus": world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"examples/demo.yml scene (4 shapes), pathtracer n={NUM_OF_RAYS} d={MAX_DEPTH} "
                                   f"rr={RR_LIMIT}, {WIDTH}x{HEIGHT}, aa={AA}, seeds {INIT_STATE}/{INIT_SEQ}",
                       "parallelism": (f"row tiles ({args.tile_rows} rows) of every frame round-robin over {world_size} "
                                       "GPU(s), " + ("peer-copy gather (HIP IPC)" if remote is not None else "RCCL gather") + " to rank 0")
                                      if world_size > 1 else "1 GPU, whole frames",
                       "frames_per_launch": B, "launches_in_flight": F,
                       "schedule": "onepass" if onepass else "passes",
                       "rays_per_frame": total_rays, "rays_per_s": total_rays * args.steps / dt},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
                         "traffic": TRAFFIC_BYTES_PER_FRAME * frames_per_launch if TRAFFIC_BYTES_PER_FRAME else None,
                         "kernel": "k_pass", "launch_ms": launch_ms, "launch_latency_ms": latency_ms,
                         "rays_per_launch": my_rays * frames_per_launch, "flops_per_ray": fpr,
                         "peak_strict_nofma": PEAK_STRICT_NOFMA_TFLOPS,
                         "frac_strict_nofma": achieved / PEAK_STRICT_NOFMA_TFLOPS},
        }
        if world_size == 1 and not args.no_cpu_baseline:
            threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
            out["cpu_baseline"] = cpu_baseline(flat, cam, rr, threads)
        print(json.dumps(out), flush=True)
    if world_size > 1:
        if remote is not None:
            # consumers let go of rank 0's buffers before rank 0 frees them
            if rank != 0:
                remote = None
            torch.cuda.synchronize()
            dist.barrier(group=cpu_group)
        dist.destroy_process_group()


def dev_n_spheres(flat):
    return sum(1 for i in range(flat.n_shapes) if flat.shapes[i].kind == 0)


def dev_n_planes(flat):
    return sum(1 for i in range(flat.n_shapes) if flat.shapes[i].kind == 1)


if __name__ == "__main__":
    main()
