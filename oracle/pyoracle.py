"""Pure-Python restatement of the reference transform (small sizes only).

TEST INFRASTRUCTURE ONLY.  A third, independent statement of what
``tests/ntt-reference.hpp`` computes, written from its *definition* rather than
its loop structure, so that the C port and the reference build are both checked
against something that shares no code with them:

    forward:  dst[j] = X[bitrev_log2m(j)],  X[k] = sum_i src[i] * w^(i*k) mod N,
              w = g^((N-1)/m)                       (ntt-reference.hpp:43-61)
    inverse:  the exact inverse map, incl. the m^{-1} scaling (:63-83)
"""
from __future__ import annotations


def bitrev(x: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def dft_forward(src: list[int], N: int, g: int) -> list[int]:
    m = len(src)
    assert m & (m - 1) == 0 and m > 0
    bits = m.bit_length() - 1
    w = pow(g, (N - 1) // m, N)
    X = [sum(src[i] * pow(w, i * k, N) for i in range(m)) % N for k in range(m)]
    return [X[bitrev(j, bits)] for j in range(m)]


def dft_inverse(src: list[int], N: int, g: int) -> list[int]:
    m = len(src)
    assert m & (m - 1) == 0 and m > 0
    bits = m.bit_length() - 1
    winv = pow(pow(g, (N - 1) // m, N), N - 2, N)
    minv = pow(m, N - 2, N)
    X = [0] * m
    for j in range(m):
        X[bitrev(j, bits)] = src[j]
    return [sum(X[k] * pow(winv, i * k, N) for k in range(m)) * minv % N for i in range(m)]


def montgomery_inverse(N: int) -> int:
    return pow(N, -1, 1 << 64)
