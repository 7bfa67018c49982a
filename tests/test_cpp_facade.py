"""The C++ facade (include/sventt/*.hpp): user code written against the reference's
template API compiles against this repository's headers, links with
libsventt_hip.so, and -- on a GPU -- reproduces the oracle bit for bit
(tests/cpp/drop_in.cpp mirrors the reference's tests/bench-ntt.cpp harness)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "drop_in")


def _build():
    import oracle
    oracle.build()
    from sve_ntt_amd import build as hip_build
    hip_build.build()
    cmd = ["g++", "-std=c++20", "-O2", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "drop_in.cpp"),
           "-L" + os.path.join(ROOT, "sve_ntt_amd"), "-lsventt_hip",
           "-L" + os.path.join(ROOT, "oracle"), "-lntt_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "sve_ntt_amd"),
           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "warning" not in r.stderr, r.stderr[-3000:]


DEV_EXE = os.path.join(ROOT, "tests", "cpp", "device_convolution")


def _build_device_harness():
    """C++ host code on hipMalloc'd buffers (HIP runtime for memory and events only)."""
    import oracle
    oracle.build()
    from sve_ntt_amd import build as hip_build
    hip_build.build()
    cmd = ["g++", "-std=c++20", "-O2", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           os.path.join(ROOT, "tests", "cpp", "device_convolution.cpp"),
           "-L" + os.path.join(ROOT, "sve_ntt_amd"), "-lsventt_hip",
           "-L" + os.path.join(ROOT, "oracle"), "-lntt_oracle", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROOT, "sve_ntt_amd"),
           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,/opt/rocm/lib", "-o", DEV_EXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_device_pointer_harness_compiles_and_links():
    _build_device_harness()
    r = subprocess.run([DEV_EXE, "--compile-only-check"], capture_output=True, text=True)
    assert r.returncode == 0 and "compiled and linked" in r.stdout


@pytest.mark.gpu
def test_device_pointer_convolution_in_cpp():
    """tests/cpp/device_convolution.cpp: cyclic convolutions on device pointers == oracle;
    N = 2^24 forward/inverse closed forms; prints the C++-side timing of the BASELINE config."""
    _build_device_harness()
    r = subprocess.run([DEV_EXE], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ALL OK" in r.stdout and "MISMATCH" not in r.stdout


SHARD_EXE = os.path.join(ROOT, "tests", "cpp", "sharded_driver")


def _build_sharded_harness():
    import oracle
    oracle.build()
    from sve_ntt_amd import build as hip_build
    hip_build.build()
    cmd = ["g++", "-std=c++20", "-O2", "-Wall", "-Wextra", "-pthread", "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           os.path.join(ROOT, "tests", "cpp", "sharded_driver.cpp"),
           "-L" + os.path.join(ROOT, "sve_ntt_amd"), "-lsventt_hip",
           "-L" + os.path.join(ROOT, "oracle"), "-lntt_oracle", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROOT, "sve_ntt_amd"),
           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,/opt/rocm/lib", "-o", SHARD_EXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_sharded_c_driver_compiles_and_links():
    _build_sharded_harness()
    r = subprocess.run([SHARD_EXE, "--compile-only-check"], capture_output=True, text=True)
    assert r.returncode == 0 and "compiled and linked" in r.stdout


@pytest.mark.gpu
def test_sharded_c_driver_with_loopback_transport():
    """sventt_sharded_forward/inverse_transport from C++: 2 and 4 ranks as host threads on the
    one GPU, exchange through a loopback transport, every rank's output == its oracle slice."""
    _build_sharded_harness()
    r = subprocess.run([SHARD_EXE], capture_output=True, text=True, timeout=900)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ALL OK" in r.stdout and "MISMATCH" not in r.stdout


@pytest.mark.gpu
def test_sharded_c_driver_config5_own_shape():
    """BASELINE configs[4] in its own shape on one GPU: N = 2^30, 8 thread-ranks x 5 buffers x 1 GiB,
    R = 2^11, exchange pipelined in 4 chunks through the loopback transport -- the plans, chunking and
    event graph of an 8-GPU node with only the wire replaced; closed-form check of sampled outputs of
    every rank, inverse == input bit for bit (tests/cpp/sharded_driver.cpp --config5)."""
    _build_sharded_harness()
    r = subprocess.run([SHARD_EXE, "--config5"], capture_output=True, text=True, timeout=1100)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ALL OK" in r.stdout and "MISMATCH" not in r.stdout
    assert "8 ranks, n=2^30" in r.stdout and "(2 row-phase passes)" in r.stdout and "two-level" in r.stdout


RCCL_EXE = os.path.join(ROOT, "tests", "cpp", "rccl_one_rank")


def _build_rccl_harness():
    import oracle
    oracle.build()
    from sve_ntt_amd import build as hip_build
    hip_build.build()
    cmd = ["g++", "-std=c++20", "-O2", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           os.path.join(ROOT, "tests", "cpp", "rccl_one_rank.cpp"),
           "-L" + os.path.join(ROOT, "sve_ntt_amd"), "-lsventt_hip",
           "-L" + os.path.join(ROOT, "oracle"), "-lntt_oracle", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl",
           "-Wl,-rpath," + os.path.join(ROOT, "sve_ntt_amd"),
           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,/opt/rocm/lib", "-o", RCCL_EXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_rccl_one_rank_harness_compiles_and_links():
    _build_rccl_harness()
    r = subprocess.run([RCCL_EXE, "--compile-only-check"], capture_output=True, text=True)
    assert r.returncode == 0 and "compiled and linked" in r.stdout


@pytest.mark.gpu
def test_sharded_c_driver_over_real_rccl_one_rank():
    """sventt_sharded_forward/_inverse with an ncclComm_t: the library's own RCCL exchange (dlopen'ed
    librccl, group of ncclSend/ncclRecv on the plan's communication stream) executed on hardware with
    a one-rank communicator -- all a one-GPU box allows -- and checked against the oracle."""
    _build_rccl_harness()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([RCCL_EXE], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ALL OK" in r.stdout and "MISMATCH" not in r.stdout


def test_facade_compiles_and_links():
    _build()
    r = subprocess.run([EXE, "--compile-only-check"], capture_output=True, text=True)
    assert r.returncode == 0 and "compiled and linked" in r.stdout


@pytest.mark.gpu
def test_facade_matches_oracle_on_gpu():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ALL OK" in r.stdout and "MISMATCH" not in r.stdout
    assert r.stdout.count("ok ") >= 10
    # the explicit 2^8 x 2^9 split of the README example reaches the planner
    assert "col 2^8" in r.stdout and "row 2^9" in r.stdout
