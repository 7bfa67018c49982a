#!/usr/bin/env python3
"""Generates tests/golden/magic_series.json: exact (big-integer) known answers for the
caller-level test tests/test_magic_series.py.

  * counts[m]  = number of magic series of order m (OEIS A052456) = the coefficient of
    q^(m^2 (m-1)/2) of the Gaussian binomial [m^2 choose m]_q, computed here in exact
    integer arithmetic through the chain [N-k+j choose j]_q, j = 1..k (every member is
    a polynomial, so each division by (1 - q^j) is exact);
  * qpochhammer[k] = coefficient list of prod_{i=1..k} (1 - q^i);
  * restricted_partitions[k] = p(0, k), p(1, k), ...: the coefficients of its reciprocal.

The reference holds the same numbers as known answers
(examples/magic-series/test-magic-series.cpp:47-77, :104-143 and :315-330); when /root/reference
is present the script checks that every decimal string it produces occurs there.
Run:  python tests/golden/make_magic_series.py      (about two minutes for m = 100)
"""
import json
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORDERS = (10, 25, 35, 42, 100)
POCHHAMMER_K = (7, 8, 10, 21)
# the moduli the reference's caller is tested with (test-magic-series.cpp:22-39) + BASELINE's
MODULI = [
    ("goldilocks, smallest generator", 0xFFFFFFFF00000001, 7),
    ("goldilocks, random generator", 0xFFFFFFFF00000001, 0xF44872F5EC1C4CC0),
    ("64-bit", 0xA3B25F400C7A8001, 5),
    ("63-bit", 0x41D33D0D1FBF8001, 6),
    ("62-bit", 0x3164C5D59B090001, 13),
    ("61-bit", 0x1E4A0E19E4548001, 3),
    ("60-bit", 0x08AA90297F870001, 3),
    ("fermat 2^16+1", 0x10001, 3),
    ("baseline", 0xFFFFFC6E80000001, 3),
]


def magic_series_exact(m: int) -> int:
    n, k = m * m, m
    d = m * m * (m - 1) // 2
    c = np.zeros(d + 1, dtype=object)
    c[0] = 1
    for j in range(1, k + 1):
        e = n - k + j
        if e <= d:
            c[e:] = c[e:] - c[:d + 1 - e].copy()  # times (1 - q^e)
        # divided by (1 - q^j): c[t] += c[t - j], in order of t == a cumulative sum per residue class
        for r in range(j):
            c[r::j] = np.cumsum(c[r::j])
    return int(c[d])


def qpochhammer(k: int) -> list[int]:
    c = [0] * (k * (k + 1) // 2 + 1)
    c[0] = 1
    for i in range(1, k + 1):
        for t in range(len(c) - 1, i - 1, -1):
            c[t] -= c[t - i]
    return c


def restricted_partitions(k: int, terms: int) -> list[int]:
    """p(0, k), ..., p(terms - 1, k): partitions of i into parts <= k = coefficients of
    1 / prod_{j=1..k} (1 - q^j)."""
    c = [0] * terms
    c[0] = 1
    for j in range(1, min(k, terms - 1) + 1):
        for t in range(j, terms):
            c[t] += c[t - j]
    return c


PARTITION_CASES = ((7, 107), (42, 45), (1234, 128))  # (k, number of terms)


def main():
    counts = {str(m): str(magic_series_exact(m)) for m in ORDERS}
    poch = {str(k): qpochhammer(k) for k in POCHHAMMER_K}
    parts = {str(k): restricted_partitions(k, n) for k, n in PARTITION_CASES}
    checked = False
    ref = "/root/reference/examples/magic-series/test-magic-series.cpp"
    if os.path.exists(ref):
        text = re.sub(r'"\s*\n\s*"', "", open(ref).read())  # join split string literals
        for m, v in counts.items():
            assert f'"{v}"' in text, f"count for m={m} disagrees with the reference's KAT"
        flat = re.sub(r"\s+", "", text)
        for k, v in poch.items():
            assert "{" + ",".join(str(x) for x in v) + "}" in flat, f"qpochhammer k={k} disagrees"
        for k, v in parts.items():
            # the reference lists at least the first 45 terms for each of these k (:104-143)
            assert "{" + ",".join(str(x) for x in v[:45]) in flat, f"partitions k={k} disagree"
        checked = True
    out = {
        "source": "tests/golden/make_magic_series.py (exact integer arithmetic)",
        "agrees_with_reference_kats": checked,
        "reference_kats": "examples/magic-series/test-magic-series.cpp:47-77,315-330",
        "ntt_length": 1 << 15,
        "moduli": [{"name": nm, "modulus": f"{p:#x}", "generator": f"{g:#x}"} for nm, p, g in MODULI],
        "counts": counts,
        "qpochhammer": poch,
        "restricted_partitions": parts,
    }
    with open(os.path.join(HERE, "magic_series.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote magic_series.json; checked against reference:", checked)


if __name__ == "__main__":
    main()
