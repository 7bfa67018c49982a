"""GPU parity: the HIP path (through the C ABI) against the oracle and the
golden vectors the real reference produced.  Bit-exact (integer work).

Mirrors the reference's integration test, tests/bench-ntt.cpp:17-65: fill the
destination with 0x55.., transform out of place, compare every element with
NTTReference.
"""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

P, G = oracle.BASELINE_P, oracle.BASELINE_G


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    import sve_ntt_amd
    return sve_ntt_amd


def dev(a: np.ndarray):
    return torch.from_numpy(a.view(np.int64)).cuda()


def host(t) -> np.ndarray:
    return t.cpu().numpy().view(np.uint64)


def _h(lst):
    return np.array([int(x, 16) for x in lst], dtype=np.uint64)


def run_forward(eng, src, n, batch=1, n0_log2=0, N=P, g=G, in_place=False):
    ntt = eng.NTT(eng.Modulus(N, g), n, n0_log2=n0_log2, batch=batch, enable_inverse=False)
    s = dev(src)
    if in_place:
        ntt.compute_forward(s)
        return host(s)
    d = torch.full_like(s, 0x5555555555555555)  # tests/bench-ntt.cpp:34
    ntt.compute_forward(d, s)
    assert np.array_equal(host(s), src), "source was modified by an out-of-place transform"
    return host(d)


def run_inverse(eng, src, n, batch=1, n0_log2=0, N=P, g=G, in_place=False):
    ntt = eng.NTT(eng.Modulus(N, g), n, n0_log2=n0_log2, batch=batch, enable_forward=False)
    s = dev(src)
    if in_place:
        ntt.compute_inverse(s)
        return host(s)
    d = torch.full_like(s, 0x5555555555555555)
    ntt.compute_inverse(d, s)
    return host(d)


def test_golden_full_vectors(eng, golden_full):
    """Every full vector the reference generated (all three primes, edge inputs)."""
    for c in golden_full:
        N, g, m = int(c["N"], 16), c["g"], 1 << c["log2m"]
        src = _h(c["src"])
        assert np.array_equal(run_forward(eng, src, m, N=N, g=g), _h(c["forward"])), (c["prime"], m, c["input"])
        assert np.array_equal(run_inverse(eng, src, m, N=N, g=g), _h(c["inverse"])), (c["prime"], m, c["input"])


@pytest.mark.parametrize("log2m", range(1, 21))
def test_every_length_vs_oracle(eng, port, log2m):
    m = 1 << log2m
    src = port.fill_splitmix(m, 1000 + log2m, P)
    want = port.forward(src, P, G)
    assert np.array_equal(run_forward(eng, src, m), want)
    assert np.array_equal(run_forward(eng, src, m, in_place=True), want)
    assert np.array_equal(run_inverse(eng, want, m), src)
    assert np.array_equal(run_inverse(eng, want, m, in_place=True), src)
    # the oracle's own inverse on arbitrary (not-a-spectrum) input
    assert np.array_equal(run_inverse(eng, src, m), port.inverse(src, P, G))


def test_golden_digests(eng, port, golden_digests):
    """Large fixtures: recipe -> transform -> digest (incl. BASELINE's 2^24)."""
    for c in golden_digests:
        N, g, m = int(c["N"], 16), c["g"], 1 << c["log2m"]
        if c["input"]["kind"] == "iota":
            src = port.fill_iota(m, int(c["input"]["start"], 16))
        else:
            src = port.fill_splitmix(m, c["input"]["seed"], N)
        fwd = run_forward(eng, src, m, N=N, g=g)
        assert [f"{x:016x}" for x in port.digest(fwd)] == c["forward_digest"], (c["prime"], c["log2m"])
        assert [f"{int(x):016x}" for x in fwd[:8]] == c["forward_head"]
        if "inverse_digest" in c:
            inv = run_inverse(eng, src, m, N=N, g=g)
            assert [f"{x:016x}" for x in port.digest(inv)] == c["inverse_digest"]


def test_readme_config_2p17_as_2p8_x_2p9(eng, port):
    """BASELINE config #2 / README.md:30-32 of the reference: n0 = 2^8, n1 = 2^9."""
    m = 1 << 17
    src = port.fill_iota(m, oracle.INPUT_I1_START)
    ntt = eng.NTT(eng.Modulus(P, G), m, n0_log2=8)
    assert "col 2^8" in ntt.describe() and "row 2^9" in ntt.describe()
    want = port.forward(src, P, G)
    buf = dev(src)
    ntt.compute_forward(buf)  # in place, as README.md:81
    assert np.array_equal(host(buf), want)
    ntt.compute_inverse(buf)
    assert np.array_equal(host(buf), src)


@pytest.mark.parametrize("log2m,n0", [(14, 1), (14, 5), (16, 3), (16, 8), (18, 7), (20, 9),
                                      (20, 11), (22, 10), (23, 11), (24, 11)])
def test_explicit_splits(eng, port, log2m, n0):
    m = 1 << log2m
    src = port.fill_splitmix(m, 5 * log2m + n0, P)
    want = port.forward(src, P, G)
    assert np.array_equal(run_forward(eng, src, m, n0_log2=n0), want)
    assert np.array_equal(run_inverse(eng, want, m, n0_log2=n0), src)


def test_full_size_2p24_roundtrip_and_oracle(eng, port):
    """BASELINE config #3: N = 2^24 forward + inverse round trip, forward == oracle."""
    m = 1 << 24
    src = port.fill_splitmix(m, 42, P)
    ntt = eng.NTT(eng.Modulus(P, G), m)
    s = dev(src)
    d = torch.full_like(s, 0x5555555555555555)
    ntt.compute_forward(d, s)
    fwd = host(d)
    assert np.array_equal(fwd, port.forward(src, P, G))
    # closed forms of tests/test-ntt-reference.cpp:45-63
    total = (int((src >> np.uint64(32)).sum(dtype=np.uint64)) << 32) + int(
        (src & np.uint64(0xFFFFFFFF)).sum(dtype=np.uint64))
    assert int(fwd[0]) == total % P
    ntt.compute_inverse(d)
    assert np.array_equal(host(d), src)


@pytest.mark.parametrize("log2m", [25, 26, 27])
def test_three_pass_sizes_properties(eng, port, log2m):
    """Beyond two passes: size-independent checks (closed forms + round trip + linearity)."""
    m = 1 << log2m
    rng = np.random.default_rng(log2m)
    a = rng.integers(0, P, size=m, dtype=np.uint64)
    ntt = eng.NTT(eng.Modulus(P, G), m)
    assert ntt.num_passes() == (2 if log2m == 25 else 3)
    da = dev(a)
    fa = torch.empty_like(da)
    ntt.compute_forward(fa, da)
    f = host(fa)
    def isum(v):  # exact integer sum of a uint64 vector (< 2^32 terms)
        return (int((v >> np.uint64(32)).sum(dtype=np.uint64)) << 32) + int(
            (v & np.uint64(0xFFFFFFFF)).sum(dtype=np.uint64))

    assert int(f[0]) == isum(a) % P
    assert int(f[1]) == (isum(a[0::2]) - isum(a[1::2])) % P
    ntt.compute_inverse(fa)
    assert np.array_equal(host(fa), a)


@pytest.mark.parametrize("log2m", [24, 28, 30])
def test_iota_closed_form_up_to_2p30(eng, log2m):
    """Sizes no CPU oracle finishes in seconds (BASELINE config #5's 2^30 on one GPU): the
    harness input of the reference, a[i] = s + i (tests/bench-ntt.cpp:31-33), has a closed form
    for EVERY output -- X[0] = m s + m(m-1)/2 and X[k] = m / (w^k - 1) for k != 0 (the
    derivative of the geometric sum) -- so sampled outputs are checked exactly; then the
    inverse must give the iota back (compared on the device)."""
    m = 1 << log2m
    s0 = oracle.INPUT_I1_START
    src = torch.arange(s0, s0 + m, dtype=torch.int64, device="cuda")
    dst = torch.full_like(src, 0x5555555555555555)
    ntt = eng.NTT(eng.Modulus(P, G), m)
    ntt.compute_forward(dst, src)
    w = pow(G, (P - 1) // m, P)
    rng = np.random.default_rng(log2m)
    where = np.unique(np.concatenate([np.arange(64), m - 1 - np.arange(64),
                                      rng.integers(0, m, size=2048)]))
    got = dst[torch.from_numpy(where).cuda()].cpu().numpy().view(np.uint64)
    for j, x in zip(where.tolist(), got.tolist()):
        k = int(format(j, f"0{log2m}b")[::-1], 2)  # output j holds frequency bitrev(j)
        want = (m * s0 + m * (m - 1) // 2) % P if k == 0 else m * pow(pow(w, k, P) - 1, -1, P) % P
        assert x == want, (j, k)
    del src
    ntt.compute_inverse(dst)
    assert torch.equal(dst, torch.arange(s0, s0 + m, dtype=torch.int64, device="cuda"))


def test_batched_2p12_full_size_sampled(eng, port):
    """BASELINE config #4 at full size: 2^16 independent N = 2^12 transforms (2 GiB, in place);
    97 sampled batches against the oracle, the whole buffer by round trip."""
    m, batch = 1 << 12, 1 << 16
    src = torch.randint(0, 1 << 62, (m * batch,), dtype=torch.int64, device="cuda")  # all < p
    buf = src.clone()
    ntt = eng.NTT(eng.Modulus(P, G), m, batch=batch)
    ntt.compute_forward(buf)
    for b in [0, 1, batch - 1] + list(range(7, batch, 697)):
        a = host(src[b * m:(b + 1) * m])
        assert np.array_equal(host(buf[b * m:(b + 1) * m]), port.forward(a, P, G)), b
    ntt.compute_inverse(buf)
    assert torch.equal(buf, src)


def test_batched_2p12_sampled(eng, port):
    """BASELINE config #4 at reduced batch: 2^10 independent N = 2^12 transforms in place;
    every batch checked against the oracle."""
    m, batch = 1 << 12, 1 << 10
    src = port.fill_splitmix(m * batch, 7, P)
    got = run_forward(eng, src, m, batch=batch, in_place=True)
    for b in range(batch):
        assert np.array_equal(got[b * m:(b + 1) * m], port.forward(src[b * m:(b + 1) * m], P, G)), b
    back = run_inverse(eng, got, m, batch=batch)
    assert np.array_equal(back, src)


@pytest.mark.parametrize("m,batch", [(2, 3), (8, 5), (16, 257), (64, 33), (1 << 9, 7), (1 << 13, 3),
                                     (1 << 15, 3), (1 << 17, 2)])
def test_ragged_batches(eng, port, m, batch):
    """Batch counts that do not fill the last tile (masking) and multi-pass batches."""
    src = port.fill_splitmix(m * batch, m + batch, P)
    got = run_forward(eng, src, m, batch=batch)
    for b in range(batch):
        assert np.array_equal(got[b * m:(b + 1) * m], port.forward(src[b * m:(b + 1) * m], P, G))
    assert np.array_equal(run_inverse(eng, got, m, batch=batch), src)


@pytest.mark.parametrize("N,g", [(oracle.TEST62_P, 3), (oracle.GOLDILOCKS_P, 7),
                                 (0x0C40000000000001, 5), (0x0003F00000000001, 11)])
def test_other_moduli(eng, port, N, g):
    """The reference's test primes (tests/ntt-tests/*.hpp:4-5, test-ntt-reference.cpp:17-23)."""
    for log2m in (3, 10, 13, 15, 18):
        m = 1 << log2m
        src = port.fill_splitmix(m, log2m, N)
        want = port.forward(src, N, g)
        assert np.array_equal(run_forward(eng, src, m, N=N, g=g), want)
        assert np.array_equal(run_inverse(eng, want, m, N=N, g=g), src)


@pytest.mark.parametrize("arith,N,g", [("auto", oracle.GOLDILOCKS_P, 7), ("generic", oracle.GOLDILOCKS_P, 7),
                                       ("fixed_point", oracle.TEST62_P, 3), ("fixed_point", 0x7FFFFFFFF9000001, None),
                                       ("fixed_point", 65537, 3)])
@pytest.mark.parametrize("log2m,n0,batch", [(1, 0, 3), (4, 0, 1), (9, 0, 5), (12, 0, 2), (13, 0, 1), (14, 3, 1),
                                            (16, 0, 1), (17, 8, 1), (20, 9, 1), (22, 0, 1)])
def test_arithmetic_back_ends(eng, port, arith, N, g, log2m, n0, batch):
    """the Goldilocks kernels (chosen automatically for
    p = 2^64 - 2^32 + 1) and the FixedPoint64 (Shoup) kernels give the oracle's results bit for
    bit, forward and inverse, on every tile family the planner reaches."""
    if g is None:  # a 63-bit prime with 2^24 | p - 1 (the largest moduli the FixedPoint64 back end takes): find a generator of the 2-power subgroup
        g = next(c for c in range(2, 200) if pow(c, (N - 1) // 2, N) == N - 1)
    m = 1 << log2m
    if (N - 1) % m:
        pytest.skip("the field has no root of this order")
    # (the oracle's splitmix fill rejects values >= N: hopeless for a 17-bit modulus)
    src = (port.fill_splitmix(m * batch, 31 + log2m, N) if N >> 32 else
           np.random.default_rng(log2m).integers(0, N, size=m * batch, dtype=np.uint64))
    ntt = eng.NTT(eng.Modulus(N, g), m, n0_log2=n0, batch=batch, arithmetic=arith)
    tag = {"auto": "[goldilocks]", "fixed_point": "[fixed-point]", "generic": ""}[arith]
    assert ntt.describe().startswith(tag) and (tag or not ntt.describe().startswith("["))
    d = dev(src)
    ntt.compute_forward(d)
    got = host(d)
    for b in range(batch):
        assert np.array_equal(got[b * m:(b + 1) * m], port.forward(src[b * m:(b + 1) * m], N, g)), (arith, b)
    ntt.compute_inverse(d)
    assert np.array_equal(host(d), src)


def test_arithmetic_back_ends_extreme_values(eng, port):
    """all-(p-1), all-zero and single-spike inputs through the special back ends"""
    for arith, N, g in (("auto", oracle.GOLDILOCKS_P, 7), ("fixed_point", oracle.TEST62_P, 3)):
        m = 1 << 13
        ntt = eng.NTT(eng.Modulus(N, g), m, arithmetic=arith)
        for src in (np.full(m, N - 1, dtype=np.uint64), np.zeros(m, dtype=np.uint64),
                    np.concatenate([[N - 1], np.zeros(m - 1)]).astype(np.uint64),
                    (np.arange(m, dtype=np.uint64) % np.uint64(2)) * np.uint64(N - 1)):
            d = dev(src)
            ntt.compute_forward(d)
            assert np.array_equal(host(d), port.forward(src, N, g)), arith
            ntt.compute_inverse(d)
            assert np.array_equal(host(d), src)


@pytest.mark.parametrize("arith,N,g", [("auto", oracle.GOLDILOCKS_P, 7), ("fixed_point", oracle.TEST62_P, 3)])
@pytest.mark.parametrize("log2m,batch", [(3, 2), (12, 3), (13, 1), (17, 1), (20, 1)])
def test_fused_product_on_the_other_back_ends(eng, port, arith, N, g, log2m, batch):
    """sventt_forward_multiply keeps its contract (operand in Montgomery form) whatever arithmetic the
    butterflies use, and the inverse divisor is honoured there too."""
    m = 1 << log2m
    src = port.fill_splitmix(m * batch, 3 + log2m, N)
    operand = port.fill_splitmix(m * batch, 5 + log2m, N)
    ntt = eng.NTT(eng.Modulus(N, g), m, batch=batch, arithmetic=arith, inverse_divisor=1)
    s, o = dev(src), dev(operand)
    om = torch.empty_like(o)
    ntt.to_montgomery(om, o)
    fused = torch.full_like(s, 0x5555555555555555)
    ntt.compute_forward_multiply(fused, s, om)
    for b in range(batch):
        f = port.forward(src[b * m:(b + 1) * m], N, g)
        want = np.array((f.astype(object) * operand[b * m:(b + 1) * m].astype(object)) % N, dtype=np.uint64)
        assert np.array_equal(host(fused)[b * m:(b + 1) * m], want), (arith, b)
    d = dev(src)
    ntt.compute_forward(d)
    ntt.compute_inverse(d)  # divisor 1: the unscaled inverse = m * x
    want = np.array((src.astype(object) * m) % N, dtype=np.uint64)
    assert np.array_equal(host(d), want)


def test_fixed_point_needs_a_modulus_below_2p63(eng):
    with pytest.raises(ValueError):  # SVENTT_ERR_INVALID_ARGUMENT
        eng.NTT(eng.Modulus(P, G), 1 << 10, arithmetic="fixed_point")


def test_edge_values(eng, port):
    for m in (2, 16, 1 << 12, 1 << 16):
        for src in (np.zeros(m, dtype=np.uint64), np.full(m, P - 1, dtype=np.uint64),
                    np.where(np.arange(m) % 2 == 0, P - 1, 1).astype(np.uint64)):
            want = port.forward(src, P, G)
            assert np.array_equal(run_forward(eng, src, m), want)
            assert np.array_equal(run_inverse(eng, want, m), src)


def test_length_one_and_host_pointers(eng, port):
    one = np.array([12345], dtype=np.uint64)
    assert np.array_equal(run_forward(eng, one, 1), one)
    # host (numpy) buffers are staged through the device, like NTT::compute_* on PageMemory
    m = 1 << 14
    src = port.fill_splitmix(m, 3, P)
    ntt = eng.NTT(eng.Modulus(P, G), m)
    dst = np.full(m, 0x5555555555555555, dtype=np.uint64)
    ntt.compute_forward(dst, src)
    assert np.array_equal(dst, port.forward(src, P, G))
    ntt.compute_inverse(dst)
    assert np.array_equal(dst, src)


def test_linearity_and_convolution(eng, port):
    """forward(a)+forward(b) == forward(a+b); forward -> pointwise -> inverse is the cyclic
    convolution (the only caller of the reference does exactly this,
    examples/magic-series/gaussian-polynomial.hpp:196-241)."""
    m = 1 << 16
    rng = np.random.default_rng(9)
    a = rng.integers(0, P, size=m, dtype=np.uint64)
    b = np.zeros(m, dtype=np.uint64)
    b[:5] = [3, 0, 7, 1, P - 2]
    ntt = eng.NTT(eng.Modulus(P, G), m)
    fa, fb = dev(a), dev(b)
    ntt.compute_forward(fa)
    ntt.compute_forward(fb)
    ab = (a.astype(object) + b.astype(object)) % P
    fab = dev(np.array(ab, dtype=np.uint64))
    ntt.compute_forward(fab)
    s = (host(fa).astype(object) + host(fb).astype(object)) % P
    assert np.array_equal(host(fab), np.array(s, dtype=np.uint64))
    prod = torch.empty_like(fa)
    ntt.pointwise_multiply(prod, fa, fb)
    ntt.compute_inverse(prod)
    got = host(prod)
    ao = a.astype(object)
    for k in (0, 1, 4, 77, m - 1):
        want = sum(int(b[j]) * int(ao[(k - j) % m]) for j in range(5)) % P
        assert int(got[k]) == want


@pytest.mark.parametrize("log2m,batch", [(0, 5), (3, 1), (10, 7), (12, 1), (13, 2), (17, 1), (20, 1), (22, 1)])
def test_forward_multiply_fused(eng, port, log2m, batch):
    """sventt_forward_multiply == compute_forward followed by pointwise_multiply (and both ==
    the oracle's forward times the operand); to/from_montgomery round trip."""
    m = 1 << log2m
    src = port.fill_splitmix(m * batch, 11 + log2m, P)
    operand = port.fill_splitmix(m * batch, 13 + log2m, P)
    ntt = eng.NTT(eng.Modulus(P, G), m, batch=batch)
    s, o = dev(src), dev(operand)
    om = torch.empty_like(o)
    ntt.to_montgomery(om, o)
    # every element: operand * 2^64 mod p (PAdic64SVE::to_montgomery, modmul/sve/p-adic-64.hpp:64-69)
    want_m = np.array([(int(v) << 64) % P for v in operand[:4096]], dtype=np.uint64)
    assert np.array_equal(host(om)[:want_m.size], want_m)
    assert int(host(om)[0]) == port.to_montgomery(int(operand[0]), P)
    back = torch.empty_like(o)
    ntt.from_montgomery(back, om)
    assert torch.equal(back, o)
    fused = torch.full_like(s, 0x5555555555555555)
    ntt.compute_forward_multiply(fused, s, om)
    two_step = torch.empty_like(s)
    ntt.compute_forward(two_step, s)
    ntt.pointwise_multiply(two_step, two_step, o)
    assert torch.equal(fused, two_step)
    f = port.forward(src[:m], P, G) if m > 1 else src[:1]
    want = (f.astype(object) * operand[:m].astype(object)) % P
    assert np.array_equal(host(fused)[:m], np.array(want, dtype=np.uint64))
    inplace = s.clone()
    ntt.compute_forward_multiply(inplace, None, om)  # in place
    assert torch.equal(inplace, fused)
    with pytest.raises(ValueError):
        ntt.compute_forward_multiply(fused, s, fused)  # operand aliases dst


def test_distinct_plans_on_concurrent_host_threads(eng, port):
    """Boundary contract (SURVEY.md 8b "Threading"): compute_* are re-entrant across plans.
    Four host threads, each with its own plan, stream and length, transform concurrently."""
    import threading
    shapes = [(1 << 12, 8), (1 << 16, 1), (1 << 20, 1), (1 << 13, 3)]
    results, errors = {}, []

    def work(i, m, batch):
        try:
            src = port.fill_splitmix(m * batch, 100 + i, P)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                ntt = eng.NTT(eng.Modulus(P, G), m, batch=batch)
                s = dev(src)
                d = torch.empty_like(s)
                for _ in range(20):
                    ntt.compute_forward(d, s, stream=stream)
                    ntt.compute_inverse(d, stream=stream)
                    ntt.compute_forward(d, stream=stream)
                stream.synchronize()
            results[i] = (src, host(d), m, batch)
        except Exception as exc:  # noqa: BLE001 - reported below
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=work, args=(i, m, b)) for i, (m, b) in enumerate(shapes)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i, (src, got, m, batch) in results.items():
        for b in range(batch):
            assert np.array_equal(got[b * m:(b + 1) * m], port.forward(src[b * m:(b + 1) * m], P, G)), (i, b)


def test_one_plan_shared_by_host_threads_with_host_pointers(eng, port):
    """ADVICE r01 / include/sventt_hip.h "Threads and devices": host-pointer calls on ONE plan
    stage through its single device buffer and must take turns, not corrupt each other.  The C++
    facade shares one plan per kernel_type process-wide (plan_handle.hpp: shared_plan), so
    independent callers of kernel_type::compute_forward do exactly this."""
    import threading
    m = 1 << 16
    ntt = eng.NTT(eng.Modulus(P, G), m)
    inputs = [port.fill_splitmix(m, 500 + i, P) for i in range(4)]
    want = [port.forward(x, P, G) for x in inputs]
    errors = []

    def work(i):
        try:
            for _ in range(12):
                out = np.full(m, 0x5555555555555555, dtype=np.uint64)
                ntt.compute_forward(out, inputs[i].copy())  # numpy arrays = host pointers
                if not np.array_equal(out, want[i]):
                    errors.append((i, "forward"))
                    return
                ntt.compute_inverse(out)
                if not np.array_equal(out, inputs[i]):
                    errors.append((i, "inverse"))
                    return
        except Exception as exc:  # noqa: BLE001 - reported below
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("log2m", [0, 1, 9, 13, 17, 24])
def test_inverse_divisor(eng, port, log2m):
    """sventt_plan_create_ex: the inverse multiplies by divisor^-1 (0 -> m, 1 -> unscaled).  The
    reference's layers scale only where an inverse_factor says so (layer/sve/radix-two.hpp:208-235):
    README-shaped kernels return m times the oracle's inverse."""
    m = 1 << log2m
    src = port.fill_splitmix(m, 77 + log2m, P)
    want = port.inverse(src, P, G) if log2m <= 20 else None
    if want is None:  # 2^24: check sampled elements against the scaled result of the default plan
        base = eng.NTT(eng.Modulus(P, G), m, enable_forward=False)
        t = dev(src)
        base.compute_inverse(t)
        want = host(t)
    for divisor, factor in ((1, m % P), (0, 1), (12345, m * pow(12345, -1, P) % P)):
        ntt = eng.NTT(eng.Modulus(P, G), m, enable_forward=False, inverse_divisor=divisor)
        d = dev(src)
        ntt.compute_inverse(d)
        got = host(d)
        idx = np.arange(m) if m <= (1 << 17) else np.random.default_rng(3).integers(0, m, 4096)
        exp = np.array([int(want[i]) * factor % P for i in idx], dtype=np.uint64)
        assert np.array_equal(got[idx], exp), (log2m, divisor)


def test_device_pointers_flag(eng, port):
    """SVENTT_DEVICE_POINTERS: same results without the pointer-kind queries."""
    m = 1 << 12
    src = port.fill_splitmix(m, 5, P)
    ntt = eng.NTT(eng.Modulus(P, G), m, device_pointers=True)
    s = dev(src)
    d = torch.empty_like(s)
    ntt.compute_forward(d, s)
    assert np.array_equal(host(d), port.forward(src, P, G))
    prod = torch.empty_like(s)
    ntt.pointwise_multiply(prod, d, d)
    ntt.compute_inverse(d)
    assert np.array_equal(host(d), src)
    assert ntt._lib.sventt_plan_device(ntt._h) == torch.cuda.current_device()


def test_transforms_are_capturable_in_a_hip_graph(eng, port):
    """Launch-bound callers (many small transforms, BASELINE configs[1]) can capture the library's launches
    in a HIP graph and replay them: after a plan's first call nothing in sventt_forward/inverse
    synchronises or allocates."""
    m = 1 << 17
    ntt = eng.NTT(eng.Modulus(P, G), m, n0_log2=8, device_pointers=True)
    src_h = port.fill_splitmix(m, 21, P)
    want = port.forward(src_h, P, G)
    src = dev(src_h)
    outs = [torch.empty_like(src) for _ in range(4)]
    back = torch.empty_like(src)
    ntt.compute_forward(outs[0], src)
    ntt.compute_inverse(back, outs[0])
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for o in outs:
                ntt.compute_forward(o, src, stream=s)
            ntt.compute_inverse(back, outs[3], stream=s)
    for o in outs:
        o.zero_()
    back.zero_()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(host(o), want)
    assert np.array_equal(host(back), src_h)


def test_plan_refuses_every_launch_from_another_device(eng):
    """include/sventt_hip.h "Threads and devices" (ADVICE r02): a plan belongs to the device it was
    created on; EVERY entry point that launches on its behalf -- not only sventt_forward/inverse --
    must return SVENTT_ERR_INVALID_ARGUMENT from a thread whose current device is another one, instead
    of launching there with tables that live elsewhere.  Needs two visible devices (skips on the
    one-GPU test box; an 8-GPU node runs it)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    from sve_ntt_amd import _lib
    m = 1 << 14
    torch.cuda.set_device(0)
    ntt = eng.NTT(eng.Modulus(P, G), m, device_pointers=True)
    x = torch.zeros(m, dtype=torch.int64, device="cuda:0")
    y = torch.zeros(m, dtype=torch.int64, device="cuda:0")
    L, h = ntt._lib, ntt._h
    try:
        torch.cuda.set_device(1)
        bad = _lib.SVENTT_ERR_INVALID_ARGUMENT
        assert L.sventt_forward(h, y.data_ptr(), x.data_ptr(), None) == bad
        assert L.sventt_run_pass(h, 0, 0, y.data_ptr(), x.data_ptr(), None) == bad
        assert L.sventt_pointwise_multiply(h, y.data_ptr(), x.data_ptr(), x.data_ptr(), m, None) == bad
        assert L.sventt_to_montgomery(h, y.data_ptr(), x.data_ptr(), m, None) == bad
        assert L.sventt_from_montgomery(h, y.data_ptr(), x.data_ptr(), m, None) == bad
        assert L.sventt_forward_multiply(h, y.data_ptr(), x.data_ptr(), x.data_ptr(), None) == bad
        assert b"current device" in L.sventt_last_error()
    finally:
        torch.cuda.set_device(0)
    assert L.sventt_forward(h, y.data_ptr(), x.data_ptr(), None) == 0
    torch.cuda.synchronize()


def test_failed_host_unregister_leaves_no_sticky_error(eng, port):
    """sventt_host_register/_unregister are used best-effort (PageMemory); a failing HIP call in
    them must not leave HIP's per-thread last error behind for the next launch to report as its own
    (launch_tile returns hipGetLastError()).  Provoked the benign way: unregistering a live, valid buffer
    that was never registered."""
    from sve_ntt_amd import _lib
    L = _lib.load()
    buf = np.zeros(1 << 12, dtype=np.uint64)
    assert L.sventt_host_unregister(buf.ctypes.data) != 0                  # never registered: HIP refuses
    assert b"hipHostUnregister" in L.sventt_last_error()
    m = 1 << 12
    ntt = eng.NTT(eng.Modulus(P, G), m)
    src = port.fill_splitmix(m, 5, P)
    out = torch.empty(m, dtype=torch.int64, device="cuda")
    ntt.compute_forward(out, dev(src))                                     # must not see the stale error
    assert np.array_equal(host(out), port.forward(src, P, G))
    assert L.sventt_host_register(buf.ctypes.data, buf.nbytes) == 0        # and the pair still works
    assert L.sventt_host_unregister(buf.ctypes.data) == 0
    torch.cuda.synchronize()


def test_sharded_columns_entry_point_equals_chunked_path(eng):
    """sventt_sharded_columns (the one-call column phase a C/C++ host would use) writes exactly
    what the Python driver's single-chunk sventt_run_pass_chunk call writes, both directions."""
    import ctypes
    from sve_ntt_amd import _lib
    from sve_ntt_amd.sharded import HipShardEngine
    n, r_log2, world, rank = 1 << 20, 8, 4, 3
    e = HipShardEngine(eng.Modulus(P, G), n, r_log2, rank, world)
    src = torch.randint(0, 1 << 62, (n // world,), dtype=torch.int64, device="cuda")
    lib = _lib.load()
    for inverse in (False, True):
        a = torch.full_like(src, 0x5555555555555555)
        b = torch.full_like(src, 0x5555555555555555)
        _lib.check(lib.sventt_sharded_columns(e._cols, int(inverse), a.data_ptr(), src.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
        e.columns_chunk(inverse, b, src, 0, 1)
        assert torch.equal(a, b), inverse
    with pytest.raises(eng.SventtError):  # a sharded plan is not a whole transform
        _lib.check(lib.sventt_forward(e._cols, src.data_ptr(), src.data_ptr(), None))
    del ctypes


def test_replica_mode_shares(eng, port):
    """ReplicaNTT: rank r of 3 transforms its contiguous share of 10 transforms; together the
    shares are the batched result (no communication involved, so ranks are emulated in turn)."""
    from sve_ntt_amd.sharded import ReplicaNTT
    m, batch, world = 1 << 9, 10, 3
    src = port.fill_splitmix(m * batch, 5, P)
    got = np.empty_like(src)
    for r in range(world):
        rep = ReplicaNTT(eng.Modulus(P, G), m, batch, rank=r, world=world)
        lo, hi = rep.first * m, (rep.first + rep.count) * m
        d = dev(src[lo:hi].copy())
        rep.forward(d)
        got[lo:hi] = host(d)
    for b in range(batch):
        assert np.array_equal(got[b * m:(b + 1) * m], port.forward(src[b * m:(b + 1) * m], P, G))


def test_plan_lifecycle_returns_device_memory(eng):
    """Creating and destroying plans (device twiddle tables, staging buffer) must not leak."""
    def cycle(count):
        for i in range(count):
            ntt = eng.NTT(eng.Modulus(P, G), 1 << (10 + i % 12), batch=1 + i % 3)
            if i % 7 == 0:  # host-pointer call allocates the plan's staging buffer
                a = np.arange(ntt.get_m() * ntt.batch, dtype=np.uint64)
                ntt.compute_forward(a)
            del ntt
    cycle(20)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    cycle(300)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, (free0, free1)


def test_error_behaviour(eng):
    with pytest.raises(ValueError):  # std::invalid_argument
        eng.NTT(eng.Modulus(P, G), 12)
    with pytest.raises(ValueError):
        eng.NTT(eng.Modulus(P, G), 1 << 32)  # 2-adicity of p is 31
    with pytest.raises(ValueError):
        eng.NTT(eng.Modulus(P, G), 1 << 10, n0_log2=10)
    fwd_only = eng.NTT(eng.Modulus(P, G), 1 << 10, enable_inverse=False)
    x = torch.zeros(1 << 10, dtype=torch.int64, device="cuda")
    with pytest.raises(eng.SventtError):  # std::logic_error
        fwd_only.compute_inverse(x)


def test_registered_host_buffers(eng, port):
    """sventt_host_register / sventt_host_unregister (what the facade's PageMemory does with its
    mapping): the host-pointer path gives the oracle's result from page-locked buffers too, the
    pair can be repeated, and a null pointer is refused."""
    import ctypes
    from sve_ntt_amd import _lib
    lib = _lib.load()
    m = 1 << 18
    src = port.fill_splitmix(m, 77, P)
    want = port.forward(src, P, G)
    buf = np.empty(2 * m, dtype=np.uint64)
    buf[:m] = src
    buf[m:] = 0x5555555555555555
    ntt = eng.NTT(eng.Modulus(P, G), m)
    for _ in range(2):
        assert lib.sventt_host_register(ctypes.c_void_p(buf.ctypes.data), buf.nbytes) == 0, lib.sventt_last_error()
        try:
            ntt.compute_forward(buf[m:], buf[:m])
            assert np.array_equal(buf[m:], want)
            ntt.compute_inverse(buf[m:])
            assert np.array_equal(buf[m:], src)
        finally:
            assert lib.sventt_host_unregister(ctypes.c_void_p(buf.ctypes.data)) == 0
    assert lib.sventt_host_register(None, 64) == _lib.SVENTT_ERR_INVALID_ARGUMENT
    assert lib.sventt_host_unregister(None) == _lib.SVENTT_ERR_INVALID_ARGUMENT
