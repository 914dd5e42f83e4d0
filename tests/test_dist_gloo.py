"""world_size > 1 path on CPU: interleaved row bands + gather over gloo (RCCL on the GPU box)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_gathers_to_the_single_rank_frame(world, orc):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_dist_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert f"DIST_OK {world}" in r.stdout and f"PIPE_OK {world}" in r.stdout and f"SAMPLES_OK {world}" in r.stdout and f"BATCH_OK {world}" in r.stdout
