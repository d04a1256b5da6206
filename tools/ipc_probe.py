"""Probe for the planned peer-copy gather: can a rank write into a buffer of rank 0 through a HIP IPC
handle exchanged over torch.distributed?  Both ranks may sit on one GPU (the handle path is the same).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 tools/ipc_probe.py
"""
import os
import time

import torch
import torch.distributed as dist
from torch.multiprocessing.reductions import reduce_tensor


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    n = 16 << 20
    box = [None]
    if rank == 0:
        gathered = torch.zeros((world, n), dtype=torch.float32, device="cuda")
        box[0] = reduce_tensor(gathered)          # (rebuild function, args with the IPC handle)
    dist.broadcast_object_list(box, src=0)
    if rank != 0:
        fn, args = box[0]
        remote = fn(*args)                        # rank 0's buffer, mapped into this process
        part = torch.full((n,), float(rank), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            remote[rank].copy_(part, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"rank {rank}: remote device {remote.device}, {n * 4 / dt / 1e9:.1f} GB/s per copy of {n * 4 >> 20} MiB", flush=True)
    dist.barrier()
    if rank == 0:
        torch.cuda.synchronize()
        ok = all(bool((gathered[r] == float(r)).all()) for r in range(1, world))
        print("ipc probe:", "ok" if ok else "MISMATCH", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
