"""Multi-rank six-step (sve_ntt_amd/sharded.py).

CPU tier: world_size 2 and 4 with gloo; the local passes are replayed on the
host by tests/cpu_sim so that the driver's sharding / exchange / gather logic is
what is under test.  GPU tier (one-GPU box): two ranks sharing the card over
RCCL, local passes in the HIP kernels.
"""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world: int, env_extra: dict, timeout: int = 300) -> str:
    env = dict(os.environ)
    env.update(env_extra)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "sharded_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.parametrize("world,log2n,r_log2,chunks", [(2, 12, 4, 1), (2, 16, 6, 4), (4, 16, 8, 2),
                                                        (2, 18, 3, 4), (4, 20, 6, 8),
                                                        (2, 22, 6, 4),  # 3 row passes
                                                        (8, 18, 6, 2)])  # the driver's widest node
def test_sharded_gloo_host_replay(world, log2n, r_log2, chunks):
    out = _run(world, {"ENGINE": "sim", "LOG2N": str(log2n), "R_LOG2": str(r_log2),
                       "CHUNKS": str(chunks)})
    assert "forward=OK" in out and "inverse=OK" in out, out


@pytest.mark.parametrize("world,log2n,r_log2,chunks", [(2, 16, 6, 4), (8, 20, 8, 2)])
def test_sharded_gloo_closed_form_check(world, log2n, r_log2, chunks):
    """The verification mode for sizes no oracle run fits (BASELINE config #5, N = 2^30 on 8
    GPUs): iota input, sampled outputs against X[k] = m / (omega^k - 1), inverse == input.
    Rehearsed here on the host replay with 2 and 8 ranks."""
    out = _run(world, {"ENGINE": "sim", "LOG2N": str(log2n), "R_LOG2": str(r_log2),
                       "CHUNKS": str(chunks), "CHECK": "closed", "SAMPLES": "256"})
    assert "forward=OK" in out and "inverse=OK" in out and "closed-form" in out, out


def test_sharded_gloo_goldilocks():
    """the sharded plans pick the Goldilocks back end for p = 2^64 - 2^32 + 1 like the local ones"""
    out = _run(2, {"ENGINE": "sim", "LOG2N": "16", "R_LOG2": "6", "CHUNKS": "2", "MODULUS": "goldilocks"})
    assert "forward=OK" in out and "inverse=OK" in out, out


@pytest.mark.gpu
def test_sharded_two_ranks_goldilocks_on_one_gpu():
    out = _run(2, {"ENGINE": "hip", "LOG2N": "22", "R_LOG2": "10", "MODULUS": "goldilocks"}, timeout=600)
    assert "forward=OK" in out and "inverse=OK" in out, out


@pytest.mark.gpu
def test_sharded_two_ranks_closed_form_on_one_gpu():
    """The same mode through the HIP kernels at 2^27 (two ranks share the card; host-bounced
    exchange): what `LOG2N=30 CHECK=closed` runs on an 8-GPU node."""
    out = _run(2, {"ENGINE": "hip", "LOG2N": "27", "R_LOG2": "11", "CHECK": "closed"}, timeout=600)
    assert "forward=OK" in out and "inverse=OK" in out and "closed-form" in out, out


@pytest.mark.gpu
@pytest.mark.parametrize("log2n,r_log2", [(20, 8), (25, 11)])
def test_sharded_two_ranks_on_one_gpu(log2n, r_log2):
    out = _run(2, {"ENGINE": "hip", "LOG2N": str(log2n), "R_LOG2": str(r_log2)}, timeout=600)
    assert "forward=OK" in out and "inverse=OK" in out, out
