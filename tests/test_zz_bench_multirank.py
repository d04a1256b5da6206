"""bench.py's N-rank paths rehearsed on one GPU: 2 ranks share the card (gloo for the control plane).

weak (the c3 default): a step is 2 frames, every frame in 2 contiguous row blocks, frame f's block (rank + f) % 2 on
`rank`.  strong: every frame split by 1-row tiles.  The pixels reach rank 0's images either stored in place by every
rank's render kernel through a HIP IPC mapping (direct, the default), or as parts copied device-to-device (p2p), or
gathered by a collective (placed by rank 0 / assembled by rt_unpack_parts).
In every case rank 0's images must be bit-identical to a whole-frame render (--verify).  The collective variant runs
over gloo with host staging here (two ranks cannot share one GPU under RCCL): it exercises bench.py's control flow,
not RCCL.  Collected late on purpose: it starts child processes."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("scaling,gather", [("weak", "direct"), ("strong", "direct"), ("weak", "p2p"), ("weak", "staged-collective"),
                                            ("strong", "p2p"), ("strong", "staged-collective")])
def test_two_ranks_on_one_gpu_assemble_bit_identical_frames(scaling, gather):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--repeats", "2", "--dist-backend", "gloo", "--scaling", scaling,
           "--gather", gather if gather in ("p2p", "direct") else "rccl", "--verify"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "verify ok" in r.stderr, r.stderr[-2000:]
    if gather == "direct":    # the handle was exchanged: no silent fallback here
        assert "pixels stored in place by every rank's kernel" in r.stderr, r.stderr[-2000:]
    if gather == "p2p":
        assert ("p2p copies in place" if scaling == "weak" else "(p2p gather)") in r.stderr, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, r.stdout[:2000]      # exactly one JSON line on stdout, whatever the process groups print
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == scaling
    if scaling == "weak":
        # the weak line carries the strong definition too (the same frames split by row tiles), verified as well
        assert line["strong"]["scaling"] == "strong" and line["strong"]["value"] > 0
        assert "verify ok: strong-scaling leg" in r.stderr, r.stderr[-2000:]


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_rccl_gather_executes_with_one_rank_through_the_n_rank_path(scaling):
    """What one GPU can execute of the RCCL transport (the reference's seam: src/imagetracer.rs:112-124, pixels collected
    into one HdrImage): `--force-dist` runs bench.py's N-rank code path with world size 1 under torch.distributed.run --
    the nccl (= RCCL) process group initialises, dist.gather moves the parts between device buffers, rank 0 places the
    blocks / rt_unpack_parts assembles the tiles, and --verify compares every image with a whole-frame render."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--repeats", "1", "--force-dist", "--dist-backend", "nccl", "--gather", "rccl", "--scaling", scaling, "--verify"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "verify ok" in r.stderr, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, r.stdout[:2000]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1
    assert "RCCL gather" in line["config"]["parallelism"], line["config"]["parallelism"]
    if scaling == "weak":
        assert "RCCL" in line["strong"]["parallelism"] and "verify ok: strong-scaling leg" in r.stderr


def test_direct_stores_fall_back_to_copies_when_the_probe_fails():
    """Before any rank's kernel stores through the mapping of rank 0's images, a child process tries it
    (bench.remote_store_probe_child); when that child fails, the run must go on with device-to-device copies."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--repeats", "1", "--dist-backend", "gloo", "--verify"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280, env=dict(os.environ, RUSTRACER_BENCH_FAIL_PROBE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "remote-store probe exited 9" in r.stderr, r.stderr[-2000:]
    assert "verify ok" in r.stderr and "p2p copies in place" in r.stderr, r.stderr[-2000:]


def test_one_gpu_bench_line_carries_the_contract():
    """`python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with the contract's keys, the roofline object
    (incl. the single-call fractions beside the multi-frame one) and the CPU baseline beside it."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--repeats", "2"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "call"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["unit"] == "path samples/s" and d["dtype"] == "f32"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None or r["traffic"] > 0
    assert 0 < r["frac_single_call"] < r["frac_single_frame_resident"] < r["frac"]
    assert abs(d["value"] - 1920 * 1080 * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
