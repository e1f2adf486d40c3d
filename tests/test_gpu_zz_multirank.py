"""Two ranks sharing the one GPU of the test box (gloo between them) run the real kernels through the
multi-rank host logic: pair-sharded predict with the reduce-scatter exchange, the data-parallel loss and
the data-parallel training step, each compared with a single-rank run (tools/check_multirank.py).

The ranks are child processes of the test (never an exec of this process).  The file sorts after the other
GPU tests on purpose: the round-end run uses pytest -x, and a failure here must not hide them."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_check(nproc: int, log_name: str, **extra_env):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MCMC_CHECK_TIMEOUT=os.environ.get("MCMC_CHECK_TIMEOUT", "150"), **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "check_multirank.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):  # keep the ranks' output where a GPU session can read it back
        with open(os.path.join(out_dir, log_name), "w") as f:
            f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
    return r


@pytest.mark.gpu
def test_two_ranks_match_single_rank():
    r = _run_check(2, "multirank_test.log")
    assert r.returncode == 0 and "multirank OK: world=2" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
def test_one_rank_rccl_collectives():
    """The RCCL calls of the multi-rank path (byte all-to-all with split sizes, fp64 / int32 all-reduce, all-gather) on a
    one-rank 'nccl' group, forced through the collectives (MCMC_FORCE_COLLECTIVES=1): what a one-GPU box can run of them."""
    r = _run_check(1, "rccl_one_rank_test.log", MCMC_FORCE_COLLECTIVES="1")
    assert r.returncode == 0 and "multirank OK: world=1 backend=nccl" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
