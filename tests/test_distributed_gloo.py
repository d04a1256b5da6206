"""N > 1 bookkeeping on CPU: world_size-2 gloo ranks each produce their part of the frame(s) with the CPU checker
standing in for the GPU kernel, gather to rank 0, and rank 0 reassembles the images.

What this covers: bench.py's host-side maps (`partition_rows`: strong scaling's row tiles; `rotating_block`: weak
scaling's rotating row blocks), padded part buffers and the gather's rank order.  What it does NOT cover: rt_render's
own partition arithmetic and rt_unpack_parts -- those run on the GPU and are tested there against whole-frame renders
(tests/test_gpu_parity.py::test_row_tile_partition_and_unpack_reassemble_the_image,
tests/test_gpu_configs.py::test_rotating_row_blocks_land_in_place_without_unpack, tests/test_zz_bench_multirank.py)."""
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


def _worker_weak(rank, world, port, w, h, n_frames, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import rustracer_amd as ra
    from bench import rotating_block
    from rustracer_amd.scenes import demo_camera, demo_world
    from rustracer_amd.world import flatten_camera, flatten_world

    world_scene = demo_world()
    flat = flatten_world(world_scene)
    rr = ra.PathTracer(world_scene, ra.BLACK, 3, 2, 1).flatten()
    block = h // world
    mine = np.zeros((n_frames, block, w, 3), dtype=np.float32)
    for f in range(n_frames):          # frame f (its own camera angle): this rank renders block (rank + f) % world
        cam = flatten_camera(demo_camera(w, h, angle_deg=10.0 * f))
        k = rotating_block(rank, f, world)
        mine[f], _ = O.fire_all_rays(flat, cam, rr, w, h, rows=(k * block, (k + 1) * block), n_threads=1)
    t = torch.from_numpy(mine)
    gathered = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gather_list=gathered, dst=0)
    if rank == 0:
        images = np.zeros((n_frames, h, w, 3), dtype=np.float32)
        for r in range(world):
            for f in range(n_frames):
                k = rotating_block(r, f, world)
                images[f, k * block:(k + 1) * block] = gathered[r][f].numpy()
        np.save(out_path, images)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_rotating_blocks_assemble_every_frame(tmp_path):
    sys.path.insert(0, HERE)
    import oracle_lib as O
    import rustracer_amd as ra
    from rustracer_amd.scenes import demo_camera, demo_world
    from rustracer_amd.world import flatten_camera, flatten_world
    w, h, n_frames = 20, 8, 3
    out = str(tmp_path / "imgs.npy")
    mp.spawn(_worker_weak, args=(2, _free_port(), w, h, n_frames, out), nprocs=2, join=True)
    world_scene = demo_world()
    got = np.load(out)
    for f in range(n_frames):
        ref, _ = O.fire_all_rays(flatten_world(world_scene), flatten_camera(demo_camera(w, h, angle_deg=10.0 * f)),
                                 ra.PathTracer(world_scene, ra.BLACK, 3, 2, 1).flatten(), w, h)
        assert np.array_equal(got[f].view(np.uint32), ref.view(np.uint32)), f
