"""bench.py's N-rank path rehearsed on one GPU: 2 ranks share the card (gloo for the control plane), every frame is
split by row tiles, the parts reach rank 0 through the peer-copy gather (HIP IPC buffers + device-to-device copies +
a CPU barrier per launch) or through the collective fallback, and rank 0's assembled images must be bit-identical to
a whole-frame render (--verify).  Collected last on purpose: it starts child processes."""
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


@pytest.mark.parametrize("gather", ["p2p", "rccl"])
def test_two_ranks_on_one_gpu_assemble_bit_identical_frames(gather):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
           "--dist-backend", "gloo", "--gather", gather, "--verify"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "verify ok" in r.stderr, r.stderr[-2000:]
    if gather == "p2p":
        assert "(p2p gather)" in r.stderr, r.stderr[-2000:]      # the handles were exchanged: no silent fallback here
    assert '"n_gpus": 2' in r.stdout
