"""No-GPU tier: planner, twiddle tables and tile index arithmetic (replayed on the
host by tests/cpu_sim), the C ABI's symbol table and its argument checking.

The replay executes the SAME tile code the GPU runs (sve_ntt_amd/csrc/tile_ntt.h
compiled for the host), one workgroup / step / thread at a time, and is compared
with the oracle.  It is test infrastructure, not a product path.
"""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle
from tests import simlib

P, G = oracle.BASELINE_P, oracle.BASELINE_G
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _h(lst):
    return np.array([int(x, 16) for x in lst], dtype=np.uint64)


def test_replay_matches_golden_vectors(golden_full):
    for c in golden_full:
        N, g, m = int(c["N"], 16), c["g"], 1 << c["log2m"]
        src = _h(c["src"])
        assert np.array_equal(simlib.transform(src, N, g, m), _h(c["forward"]))
        assert np.array_equal(simlib.transform(src, N, g, m, inverse=True), _h(c["inverse"]))


@pytest.mark.parametrize("log2m", range(1, 21))
def test_replay_every_length(port, log2m):
    m = 1 << log2m
    src = port.fill_splitmix(m, 77 + log2m, P)
    want = port.forward(src, P, G)
    assert np.array_equal(simlib.transform(src, P, G, m), want)
    assert np.array_equal(simlib.transform(want, P, G, m, inverse=True), src)


@pytest.mark.parametrize("log2m,n0", [(6, 1), (8, 5), (14, 1), (14, 3), (14, 11), (17, 8), (19, 6),
                                      (20, 10), (22, 11)])
def test_replay_explicit_splits(port, log2m, n0):
    m = 1 << log2m
    src = port.fill_splitmix(m, log2m * 3 + n0, P)
    want = port.forward(src, P, G)
    assert np.array_equal(simlib.transform(src, P, G, m, n0_log2=n0), want)
    assert np.array_equal(simlib.transform(want, P, G, m, n0_log2=n0, inverse=True), src)
    # and it is the same split the oracle's six-step restatement uses
    assert np.array_equal(port.forward_sixstep(src, 1 << n0, P, G), want)


def test_replay_three_pass_plan(port):
    """2^25 still fits two passes (2^12 columns of 4), 2^26 needs col | col | row; the shapes
    through the planner only (a three-pass replay runs in tests/test_sharded.py)."""
    shape = simlib.plan_shape(P, G, 1 << 25)
    assert [(s["kind"], s["logl"]) for s in shape] == [(1, 12), (0, 13)]
    shape = simlib.plan_shape(P, G, 1 << 26)
    assert [s["kind"] for s in shape] == [1, 1, 0]
    assert sum(s["logl"] for s in shape) == 26
    shape = simlib.plan_shape(P, G, 1 << 27, inverse=True)
    assert [s["kind"] for s in shape] == [0, 1, 1] and sum(s["logl"] for s in shape) == 27


@pytest.mark.parametrize("m,batch", [(2, 3), (8, 5), (16, 257), (64, 33), (512, 9), (1 << 13, 3),
                                     (1 << 15, 2)])
def test_replay_ragged_batches(port, m, batch):
    src = port.fill_splitmix(m * batch, m + batch, P)
    got = simlib.transform(src, P, G, m, batch=batch)
    for b in range(batch):
        assert np.array_equal(got[b * m:(b + 1) * m], port.forward(src[b * m:(b + 1) * m], P, G))
    assert np.array_equal(simlib.transform(got, P, G, m, batch=batch, inverse=True), src)


@pytest.mark.parametrize("m,batch,n0", [(1, 3, 0), (2, 1, 0), (64, 5, 0), (1 << 11, 1, 0), (1 << 12, 3, 0),
                                        (1 << 13, 1, 0), (1 << 14, 1, 0), (1 << 14, 1, 3), (1 << 16, 2, 0)])
def test_replay_forward_multiply(port, m, batch, n0):
    """sventt_forward_multiply: forward transform with the pointwise product (Montgomery-form
    operand) fused into the final pass == oracle forward, then an element-wise product."""
    src = port.fill_splitmix(m * batch, m + batch, P)
    operand = port.fill_splitmix(m * batch, 3 * m + batch, P)
    op_mont = np.array([port.to_montgomery(int(x), P) for x in operand], dtype=np.uint64)
    got = simlib.forward_multiply(src, op_mont, P, G, m, n0_log2=n0, batch=batch)
    for b in range(batch):
        f = port.forward(src[b * m:(b + 1) * m], P, G)
        want = (f.astype(object) * operand[b * m:(b + 1) * m].astype(object)) % P
        assert np.array_equal(got[b * m:(b + 1) * m], np.array(want, dtype=np.uint64)), b


@pytest.mark.parametrize("N,g", [(oracle.TEST62_P, 3), (oracle.GOLDILOCKS_P, 7),
                                 (0x0C40000000000001, 5), (0x0002580000000001, 11)])
def test_replay_other_moduli(port, N, g):
    for log2m in (1, 5, 12, 14, 16):
        m = 1 << log2m
        src = port.fill_splitmix(m, log2m, N)
        want = port.forward(src, N, g)
        assert np.array_equal(simlib.transform(src, N, g, m), want)
        assert np.array_equal(simlib.transform(want, N, g, m, inverse=True), src)


def test_plan_shapes():
    # BASELINE config #2 (README.md:30-32 of the reference): 2^17 = 2^8 x 2^9
    s = simlib.plan_shape(P, G, 1 << 17)
    assert [(x["kind"], x["logl"]) for x in s] == [(1, 8), (0, 9)]
    # ... on the fine tiles (E = 4): 128 column tiles of 4 columns, one row per tile
    assert [(x["loge"], x["f0"], x["logt"], x["grid"]) for x in s] == [(2, 2, 10, 128), (2, 0, 9, 256)]
    # two-pass transforms: fine tiles up to n*batch = 2^20, the 2^12/2^13-element tiles from 2^21 on
    assert all(x["loge"] == 2 for x in simlib.plan_shape(P, G, 1 << 20))
    assert all(x["loge"] == 2 for x in simlib.plan_shape(P, G, 1 << 17, batch=8))
    assert all(x["loge"] == 4 for x in simlib.plan_shape(P, G, 1 << 21))
    assert all(x["loge"] == 4 for x in simlib.plan_shape(P, G, 1 << 17, batch=16))
    assert all(x["loge"] == 4 for x in simlib.plan_shape(P, G, 1 << 22))
    assert all(x["loge"] == 2 for x in simlib.plan_shape(P, G, 1 << 10, batch=1 << 11))
    assert all(x["loge"] == 4 for x in simlib.plan_shape(P, G, 1 << 10, batch=1 << 12))
    # 2^12 and 2^13 stay one pass on the big tiles whatever the batch
    assert [(x["kind"], x["loge"]) for x in simlib.plan_shape(P, G, 1 << 12, batch=4)] == [(0, 4)]
    # BASELINE config #3: 2^24 = 2^11 columns x 2^13 rows, 8-column tiles
    s = simlib.plan_shape(P, G, 1 << 24)
    assert [(x["kind"], x["logl"]) for x in s] == [(1, 11), (0, 13)]
    assert s[0]["f0"] in (2, 3)  # 4-column tiles by default, 8 with SVENTT_COL_SLIM=0
    assert s[0]["grid"] == (1 << 13) >> s[0]["f0"] and s[1]["grid"] == 1 << 11
    # config #4: one workgroup per N = 2^12 transform
    s = simlib.plan_shape(P, G, 1 << 12, batch=1 << 16)
    assert [(x["kind"], x["logl"], x["grid"]) for x in s] == [(0, 12, 1 << 16)]
    for logn in range(1, 32):
        for inv in (False, True):
            s = simlib.plan_shape(P, G, 1 << logn, inverse=inv)
            assert sum(x["logl"] for x in s) == logn
            assert all(x["logl"] <= (12 if x["kind"] == 1 else 13) for x in s)
    # three passes (out of the Infinity Cache): the split comes from the measured per-pass costs
    # (plan_core.h: large_pass_cost) -- a short, wide first pass (address translation), then the rest
    want = {26: [7, 7, 12], 27: [7, 8, 12], 28: [7, 8, 13], 30: [7, 11, 12], 31: [7, 11, 13]}
    for logn, lens in want.items():
        s = simlib.plan_shape(P, G, 1 << logn)
        assert [x["logl"] for x in s] == lens, (logn, s)
        assert s[0]["f0"] == 5  # 32 columns = 256-byte row segments
        back = simlib.plan_shape(P, G, 1 << logn, inverse=True)
        assert [x["logl"] for x in back] == lens[::-1]
    # two-pass plans over more than 2^26 elements prefer a 2^12 row pass (large batches, sharded row phases)
    assert [x["logl"] for x in simlib.plan_shape(P, G, 1 << 19, batch=1 << 8)] == [7, 12]
    assert [x["logl"] for x in simlib.plan_shape(P, G, 1 << 19, batch=1 << 4)] == [6, 13]


def test_planner_errors():
    with pytest.raises(simlib.SimError, match="power of two"):
        simlib.plan_shape(P, G, 12)
    with pytest.raises(simlib.SimError, match="no such root"):
        simlib.plan_shape(P, G, 1 << 32)  # 2-adicity of the BASELINE prime is 31
    with pytest.raises(simlib.SimError, match="n0_log2"):
        simlib.plan_shape(P, G, 1 << 10, n0_log2=10)
    with pytest.raises(simlib.SimError, match="n0_log2"):
        simlib.plan_shape(P, G, 1 << 20, n0_log2=13)
    with pytest.raises(simlib.SimError, match="too few columns"):
        simlib.plan_shape(P, G, 1 << 6, n0_log2=5)  # 2 columns: narrower than the narrowest tile
    with pytest.raises(simlib.SimError, match="generate"):
        simlib.plan_shape(P, 4, 1 << 10)  # 4 = 2^2 is not a generator; order-n root check


def test_device_arithmetic_restatement(port):
    """field64.h (host build) against the oracle's PAdic64 restatement."""
    L = simlib.load()
    rng = np.random.default_rng(11)
    for N in (P, oracle.GOLDILOCKS_P, oracle.TEST62_P, 0x0003F00000000001):
        assert L.sim_montgomery_inverse(N) == port.montgomery_inverse(N)
        edge = [0, 1, 2, N - 1, N - 2, (1 << 32) - 1, 1 << 32, (1 << 63) % N]
        vals = edge + [int(x) for x in rng.integers(0, N, size=300, dtype=np.uint64)]
        for a, b in zip(vals, reversed(vals)):
            assert L.sim_addmod(a, b, N) == (a + b) % N
            assert L.sim_submod(a, b, N) == (a - b) % N
            w = port.to_montgomery(b, N)
            assert L.sim_to_montgomery(b, N) == w
            assert L.sim_montmul(a, w, N) == a * b % N
            assert L.sim_montmul(a, b, N) == port.padic_multiply_normalize(
                a, b, port.padic_precompute(b, N), N)
        # the multiplicand need not be reduced (a GS butterfly feeds a raw difference)
        assert L.sim_montmul((1 << 64) - 1, port.to_montgomery(5, N), N) == ((1 << 64) - 1) * 5 % N


def test_lds_swizzle_is_a_bijection_and_conflict_free():
    """The XOR swizzle must permute the tile and keep every step's ds_read_b64/ds_write_b64
    wave access at <= 2 lanes per bank pair within each 32-lane half."""
    L = simlib.load()
    for logt in (12, 13, 14):
        n = 1 << logt
        phys = np.array([L.sim_lds_phys(i) for i in range(n)])
        assert sorted(phys) == list(range(n))
    # row tile 2^13 with steps (4,4,3,2), E = 16: thread tid, set g -> element indices
    logt, nt = 13, 512
    worst = 0
    hi = 13
    for k in (4, 4, 3, 2):
        lo = hi - k
        for v in range(1 << k):
            for g in range(16 >> k):
                for wave in range(0, nt, 64):
                    for half in (0, 32):
                        banks = {}
                        for lane in range(32):
                            s = wave + half + lane + g * nt
                            I = ((s >> lo) << hi) | (v << lo) | (s & ((1 << lo) - 1))
                            b = L.sim_lds_phys(I) & 31
                            banks[b] = banks.get(b, 0) + 1
                        worst = max(worst, max(banks.values()))
        hi = lo
    assert worst <= 2, worst


def test_set_mappings_are_bijective_and_wave_local():
    """tile_ntt.h "which thread holds which radix set": in every step of every registered tile shape the
    sets cover the tile exactly once, and in chunk-preserving steps each wave holds only its own chunk
    (which is what lets the exchange between two such steps skip the workgroup barrier)."""
    from tests import simlib
    L = simlib.load()
    assert L.sim_check_set_mappings() > 200  # tile shapes checked (negative: the first bad one)
    # the two N = 2^24 kernels: one workgroup barrier per tile (r01: three and two)
    assert L.sim_group_barriers(0, 13, 0, 0, 0, 4) == 1   # ROW 2^13 forward, Steps<4,4,3,2>
    assert L.sim_group_barriers(1, 11, 0, 1, 2, 4) == 1   # COL 2^11 x T4 forward, Steps<4,3,4>
    assert L.sim_group_barriers(0, 13, 1, 0, 0, 4) == 1 and L.sim_group_barriers(1, 11, 1, 1, 2, 4) == 1


def test_c_abi_exports_every_declared_symbol():
    from sve_ntt_amd import _lib
    lib = _lib.load()  # raises if the .so is missing: there is no fallback
    header = open(os.path.join(ROOT, "include", "sventt_hip.h")).read()
    declared = set(re.findall(r"\b(sventt_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.sventt_version().startswith(b"sventt-hip")


def test_c_abi_argument_checking_needs_no_device():
    from sve_ntt_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.sventt_plan_create(P, G, 12, 0, 1, 3, ctypes.byref(h)) == _lib.SVENTT_ERR_INVALID_ARGUMENT
    assert b"power of two" in lib.sventt_last_error()
    assert lib.sventt_plan_create(P, G, 1 << 40, 0, 1, 3, ctypes.byref(h)) == _lib.SVENTT_ERR_INVALID_ARGUMENT
    assert b"no such root" in lib.sventt_last_error()
    assert lib.sventt_plan_create(P, G, 1 << 10, 0, 0, 3, ctypes.byref(h)) == _lib.SVENTT_ERR_INVALID_ARGUMENT
    assert lib.sventt_plan_create(P, G, 1 << 10, 0, 1, 0, ctypes.byref(h)) == _lib.SVENTT_ERR_INVALID_ARGUMENT
    assert lib.sventt_sharded_plan_create(P, G, 1 << 20, 8, 3, 2, 3, ctypes.byref(h)) == _lib.SVENTT_ERR_INVALID_ARGUMENT
    assert lib.sventt_forward(None, None, None, None) == _lib.SVENTT_ERR_INVALID_ARGUMENT
    # a composite modulus would give garbage tables (Fermat inverses): refused (Miller-Rabin)
    composite = 0xFFFFFC6E80000001 - 2 ** 32  # odd, 2^10 | composite - 1, not prime
    assert (composite - 1) % 1024 == 0 and pow(2, composite - 1, composite) != 1
    assert lib.sventt_plan_create(composite, 3, 1 << 10, 0, 1, 3, ctypes.byref(h)) == _lib.SVENTT_ERR_INVALID_ARGUMENT
    assert b"prime" in lib.sventt_last_error()
    carmichael = 561  # 3 * 11 * 17: passes Fermat for every coprime base
    assert lib.sventt_plan_create(carmichael, 2, 16, 0, 1, 3, ctypes.byref(h)) == _lib.SVENTT_ERR_INVALID_ARGUMENT
    assert lib.sventt_plan_create(P, G, 1 << 10, 0, 1, 3 | 64, ctypes.byref(h)) == _lib.SVENTT_ERR_INVALID_ARGUMENT
    assert b"flag" in lib.sventt_last_error()
    # an inverse divisor that is 0 mod p has no inverse
    assert lib.sventt_plan_create_ex(97, 5, 32, 0, 1, 3, 97 * 3, ctypes.byref(h)) == _lib.SVENTT_ERR_INVALID_ARGUMENT
    assert lib.sventt_plan_device(None) == -1
    import torch
    if not torch.cuda.is_available():
        # valid arguments but no GPU: the library refuses instead of falling back
        assert lib.sventt_plan_create(P, G, 1 << 10, 0, 1, 3, ctypes.byref(h)) == _lib.SVENTT_ERR_NO_DEVICE
        import sve_ntt_amd
        with pytest.raises(sve_ntt_amd.SventtError):
            sve_ntt_amd.NTT(sve_ntt_amd.BASELINE_MODULUS, 1 << 10)


def test_product_never_touches_the_oracle():
    """The shipped package must not import, link or name the checker."""
    pkg = os.path.join(ROOT, "sve_ntt_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        if os.path.basename(dirpath) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "ntt_oracle" not in text and "libsim" not in text and "cpu_sim/" not in text.replace(
                    "tests/cpu_sim", ""), f


def test_batch_partition_covers_the_batch_once():
    """Replica mode for many small transforms (SURVEY.md 8e "small N"): contiguous shares."""
    from sve_ntt_amd.sharded import batch_partition
    for batch in (0, 1, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            shares = [batch_partition(batch, world, r) for r in range(world)]
            assert shares[0][0] == 0 and sum(c for _, c in shares) == batch
            assert all(shares[r][0] + shares[r][1] == shares[r + 1][0] for r in range(world - 1))
            assert max(c for _, c in shares) - min(c for _, c in shares) <= 1
    with pytest.raises(ValueError):
        batch_partition(4, 2, 2)
