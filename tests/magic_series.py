"""A caller-level workload on top of the transform: the number of magic series of
order m modulo a 64-bit prime, i.e. the coefficient of q^(m^2 (m-1)/2) of the Gaussian
binomial [m^2 choose m]_q.

This is what the one real caller of the reference computes
(examples/magic-series/gaussian-polynomial.hpp:147-253, KATs in
examples/magic-series/test-magic-series.cpp:299-333): forward transform ->
pointwise product -> inverse transform, over and over with ONE fixed transform
length (2^15 there and here).  The arithmetic below is an independent
formulation written for this test suite (block-wise power-series division with a
Newton-iterated reciprocal); only the call pattern on the transform is shared.

TEST-ONLY.  The transform is injected as a backend:
  * EngineBackend -- sve_ntt_amd.NTT on the GPU (forward / pointwise_multiply / inverse
    in the Newton iteration, the fused forward_multiply in the block loop, through the
    C ABI), the thing under test;
  * OracleBackend -- the CPU oracle, used by the no-GPU tier to validate this file
    itself against the same known answers.
"""
from __future__ import annotations

import numpy as np

U64 = np.uint64


def submod(a: np.ndarray, b: np.ndarray, p: int) -> np.ndarray:
    r = a - b  # wraps mod 2^64
    return np.where(a < b, r + U64(p), r)


def negmod(a: np.ndarray, p: int) -> np.ndarray:
    return np.where(a == 0, a, U64(p) - a)


def one_minus_q_powers(exponents, length: int, p: int) -> np.ndarray:
    """prod_e (1 - q^e) mod q^length, coefficients in [0, p)."""
    a = np.zeros(length, dtype=U64)
    a[0] = 1
    for e in exponents:
        if e < length:
            a[e:] = submod(a[e:], a[:length - e].copy(), p)
    return a


class OracleBackend:
    """Cyclic products through the CPU oracle (checker for this file's algebra)."""

    def __init__(self, port, p: int, g: int):
        self.port, self.p, self.g = port, p, g

    def forward(self, a: np.ndarray):
        return self.port.forward(a, self.p, self.g)

    def pointwise(self, fa, fb):
        return np.array((fa.astype(object) * fb.astype(object)) % self.p, dtype=U64)

    def inverse(self, fa) -> np.ndarray:
        return self.port.inverse(fa, self.p, self.g)

    def prepare(self, fa):
        return fa

    def forward_times(self, a: np.ndarray, prepared):
        return self.pointwise(self.forward(a), prepared)


class EngineBackend:
    """The same three operations on the GPU; spectra stay resident as torch tensors."""

    def __init__(self, eng, p: int, g: int):
        import torch
        self.torch, self.eng, self.mod = torch, eng, eng.Modulus(p, g)
        self.plans = {}

    def _plan(self, n: int):
        if n not in self.plans:
            self.plans[n] = self.eng.NTT(self.mod, n)
        return self.plans[n]

    def forward(self, a: np.ndarray):
        t = self.torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
        self._plan(a.size).compute_forward(t)
        return t

    def pointwise(self, fa, fb):
        out = self.torch.empty_like(fa)
        self._plan(fa.numel()).pointwise_multiply(out, fa, fb)
        return out

    def inverse(self, fa) -> np.ndarray:
        out = self.torch.empty_like(fa)
        self._plan(fa.numel()).compute_inverse(out, fa)
        return out.cpu().numpy().view(U64)

    def prepare(self, fa):
        """A spectrum that will multiply many transforms: to Montgomery form once, as the
        reference's caller does (gaussian-polynomial.hpp:177-179)."""
        out = self.torch.empty_like(fa)
        self._plan(fa.numel()).to_montgomery(out, fa)
        return out

    def forward_times(self, a: np.ndarray, prepared):
        """forward(a) (.) spectrum with the product fused into the transform's last pass."""
        t = self.torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
        self._plan(a.size).compute_forward_multiply(t, None, prepared)
        return t


def _padded(a: np.ndarray, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=U64)
    out[:a.size] = a
    return out


def reciprocal(be, d_poly: np.ndarray, length: int, p: int) -> np.ndarray:
    """1/d_poly mod q^length (d_poly[0] == 1, length a power of two) by Newton's
    iteration x <- x (2 - d x); every product is a cyclic product of length 4s >= 3s."""
    assert int(d_poly[0]) == 1
    x = np.ones(1, dtype=U64)
    s = 1
    while s < length:
        s2, n = 2 * s, 4 * s
        fx = be.forward(_padded(x, n))
        t = be.inverse(be.pointwise(be.forward(_padded(d_poly[:s2], n)), fx))[:s2]
        u = negmod(t, p)
        u[0] = (2 - int(t[0])) % p
        x = be.inverse(be.pointwise(be.forward(_padded(u, n)), fx))[:s2].copy()
        s = s2
    return x


def gaussian_binomial_coefficient(be, n: int, k: int, d: int, p: int, ntt_len: int = 1 << 15) -> int:
    """[q^d] of prod_{i=1..k} (1 - q^(n-k+i)) / (1 - q^i)  mod p, with transforms of
    ONE length `ntt_len` (blocks of ntt_len/2 quotient coefficients per round)."""
    if d > k * (n - k):
        raise ValueError("d is out of range")
    block = ntt_len // 2
    den = one_minus_q_powers(range(1, k + 1), k * (k + 1) // 2 + 1, p)
    if den.size > block:
        raise ValueError("NTT length is too small")
    num = one_minus_q_powers(range(n - k + 1, n + 1), d + 1, p)
    f_den = be.prepare(be.forward(_padded(den, ntt_len)))
    f_rec = be.prepare(be.forward(_padded(reciprocal(be, den, block, p), ntt_len)))
    rem = _padded(num[:block], block)
    t = 0
    while True:
        quot = be.inverse(be.forward_times(_padded(rem, ntt_len), f_rec))[:block]
        if d < (t + 1) * block:
            return int(quot[d - t * block])
        back = be.inverse(be.forward_times(_padded(quot, ntt_len), f_den))
        # quot * den reproduces the window exactly; what spills over is owed by the next one
        if not np.array_equal(back[:block], rem):
            raise AssertionError("block division lost the remainder invariant")
        t += 1
        rem = submod(_padded(num[t * block:(t + 1) * block], block), back[block:], p)


def magic_series_count(be, m: int, p: int, ntt_len: int = 1 << 15) -> int:
    return gaussian_binomial_coefficient(be, m * m, m, m * m * (m - 1) // 2, p, ntt_len)
