"""N > 1 path on CPU: world_size-2 gloo ranks each produce their row-tile part (with the CPU
checker standing in for the GPU kernel), gather to rank 0, and rank 0 reassembles the image with
the same part -> image mapping rt_unpack_parts implements.  Covers partition bookkeeping, padded
part buffers and the gather; the GPU kernels themselves are covered by -m gpu tests."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, tile_rows, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import rustracer_amd as ra
    from bench import partition_rows
    from rustracer_amd.scenes import demo_camera, demo_world
    from rustracer_amd.world import flatten_camera, flatten_world

    world_scene = demo_world()
    flat, cam = flatten_world(world_scene), flatten_camera(demo_camera(w, h))
    rr = ra.PathTracer(world_scene, ra.BLACK, 3, 2, 1).flatten()
    rows = [partition_rows(h, p, world, tile_rows) for p in range(world)]
    stride = max(len(r) for r in rows) * w
    # this rank's part: its rows in part-local order (what rt_render writes for rt_partition{rank,...})
    part = np.zeros((stride, 3), dtype=np.float32)
    for i, row in enumerate(rows[rank]):
        img, _ = O.fire_all_rays(flat, cam, rr, w, h, rows=(row, row + 1), n_threads=1)
        part[i * w:(i + 1) * w] = img[0]
    t = torch.from_numpy(part)
    gathered = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gather_list=gathered, dst=0)
    if rank == 0:
        image = np.zeros((h, w, 3), dtype=np.float32)
        for p in range(world):
            g = gathered[p].numpy()
            for i, row in enumerate(rows[p]):
                image[row] = g[i * w:(i + 1) * w]
        np.save(out_path, image)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tile_rows", [1, 4])
def test_two_rank_row_tile_gather_reassembles_the_frame(tmp_path, tile_rows):
    sys.path.insert(0, HERE)
    import oracle_lib as O
    import rustracer_amd as ra
    from rustracer_amd.scenes import demo_camera, demo_world
    from rustracer_amd.world import flatten_camera, flatten_world
    w, h = 24, 10
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(2, _free_port(), w, h, tile_rows, out), nprocs=2, join=True)
    world_scene = demo_world()
    ref, _ = O.fire_all_rays(flatten_world(world_scene), flatten_camera(demo_camera(w, h)),
                             ra.PathTracer(world_scene, ra.BLACK, 3, 2, 1).flatten(), w, h)
    assert np.array_equal(np.load(out).view(np.uint32), ref.view(np.uint32))
