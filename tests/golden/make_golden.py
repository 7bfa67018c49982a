#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (oracle/_ref).

Run in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

Everything written here is DATA: inputs (or their recipe) and the outputs that
``tests/ntt-reference.hpp`` (class NTTReference) and
``include/sventt/modulus.hpp`` (class Modulus) of the reference produced, via
``oracle/_ref/libntt_ref.so`` = those headers compiled in place.  No reference
source is stored.  The fixtures travel to the GPU box; /root/reference does not.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def hx(v) -> str:
    return f"{int(v):016x}"


def main() -> None:
    ref = oracle.reference()
    port = oracle.port()  # only for input recipes + digests (not outputs)

    primes = [
        ("baseline", oracle.BASELINE_P, oracle.BASELINE_G),
        ("test62", oracle.TEST62_P, oracle.TEST62_G),
        ("goldilocks", oracle.GOLDILOCKS_P, oracle.GOLDILOCKS_G),
    ]

    # ---- 1. full small vectors ------------------------------------------------
    full = {"_doc": "full input/output vectors; outputs from the reference's "
                    "NTTReference (tests/ntt-reference.hpp:43-83)", "cases": []}
    for name, N, g in primes:
        for log2m in (0, 1, 2, 3, 4, 6, 10):
            m = 1 << log2m
            inputs = {
                "iota1": np.arange(1, m + 1, dtype=np.uint64),
                "I1": port.fill_iota(m, oracle.INPUT_I1_START % (N - m)),
                "I2": port.fill_splitmix(m, 42, N),
            }
            if log2m <= 4:
                # edge values: 0, 1, N-1 patterns
                e = np.zeros(m, dtype=np.uint64)
                e[::2] = N - 1
                e[1::2] = 1 if m > 1 else N - 1
                inputs["edges"] = e
                inputs["allmax"] = np.full(m, N - 1, dtype=np.uint64)
            for iname, src in inputs.items():
                if log2m == 10 and name != "baseline" and iname != "I1":
                    continue
                fwd = ref.forward(src, N, g)
                inv = ref.inverse(src, N, g)
                full["cases"].append({
                    "prime": name, "N": hx(N), "g": g, "log2m": log2m,
                    "input": iname,
                    "src": [hx(x) for x in src],
                    "forward": [hx(x) for x in fwd],
                    "inverse": [hx(x) for x in inv],
                })
    with open(os.path.join(HERE, "ntt_full_vectors.json"), "w") as f:
        json.dump(full, f, indent=0, separators=(",", ":"))

    # ---- 2. digests of large transforms ----------------------------------------
    dig = {"_doc": "input recipe + digest (fnv1a64, xor, wrapping sum) of the "
                   "reference's forward/inverse outputs; recipes: I1 = start+i "
                   "(tests/bench-ntt.cpp:31-33 with the pinned start of "
                   "SURVEY.md 8d), I2 = splitmix64(seed) with rejection >= N",
           "cases": []}
    for name, N, g in primes:
        sizes = (12, 13, 15, 17, 20, 24) if name == "baseline" else (12, 13, 15, 17)
        for log2m in sizes:
            m = 1 << log2m
            for iname in ("I1", "I2"):
                if log2m >= 24 and iname == "I2":
                    continue
                if iname == "I1":
                    start = oracle.INPUT_I1_START % (N - m)
                    src = port.fill_iota(m, start)
                    recipe = {"kind": "iota", "start": hx(start)}
                else:
                    src = port.fill_splitmix(m, 42, N)
                    recipe = {"kind": "splitmix64", "seed": 42}
                t0 = time.time()
                fwd = ref.forward(src, N, g)
                t1 = time.time()
                case = {
                    "prime": name, "N": hx(N), "g": g, "log2m": log2m,
                    "input": recipe,
                    "src_digest": [hx(x) for x in port.digest(src)],
                    "forward_digest": [hx(x) for x in port.digest(fwd)],
                    "forward_head": [hx(x) for x in fwd[:8]],
                    "forward_tail": [hx(x) for x in fwd[-4:]],
                }
                if log2m <= 20:
                    inv = ref.inverse(src, N, g)
                    case["inverse_digest"] = [hx(x) for x in port.digest(inv)]
                    case["inverse_head"] = [hx(x) for x in inv[:8]]
                dig["cases"].append(case)
                print(f"{name} 2^{log2m} {iname}: forward {t1 - t0:.2f}s", flush=True)
    with open(os.path.join(HERE, "ntt_digests.json"), "w") as f:
        json.dump(dig, f, indent=1)

    # ---- 3. field constants -----------------------------------------------------
    fld = {"_doc": "Modulus<p,g> constants from the reference "
                   "(include/sventt/modulus.hpp:36-68,115-132) and PAdic64Scalar "
                   "domain conversions (modmul/scalar/p-adic-64.hpp:16-29)",
           "primes": []}
    for name, N, g in primes:
        entry = {"prime": name, "N": hx(N), "g": g,
                 "generator": ref.generator(N),
                 "montgomery_inverse": hx(ref.montgomery_inverse(N)),
                 "to_montgomery_1": hx(ref.to_montgomery(1, N)),
                 "roots": [], "bad_orders": [], "montgomery_samples": []}
        orders = [1 << k for k in (1, 2, 3, 8, 12, 17, 24, 28, 31) if (N - 1) % (1 << k) == 0]
        if name == "goldilocks":
            # tests/test-modulus.cpp:17-19
            orders += [3, 5, 17, 257, 65537, (1 << 14) * 5 * 17 * 257]
        for order in orders:
            entry["roots"].append({"order": order,
                                   "forward": hx(ref.root_forward(N, order)),
                                   "inverse": hx(ref.root_inverse(N, order))})
        for order in (7 if name != "goldilocks" else 11, 1 << 40):
            try:
                ref.root_forward(N, order)
            except ValueError:
                entry["bad_orders"].append(order)
        for b in port.fill_splitmix(8, 7, N):
            b = int(b)
            entry["montgomery_samples"].append({
                "b": hx(b), "to": hx(ref.to_montgomery(b, N)),
                "from": hx(ref.from_montgomery(b, N)),
                "precompute": hx(ref.padic_precompute(b, N))})
        fld["primes"].append(entry)
    fld["bitreverse"] = [{"x": hx(x), "r": hx(ref.bitreverse64(x))}
                         for x in (0, 1, 2, 0x8000000000000000, 0x0123456789ABCDEF,
                                   0xFFFFFFFF00000000, 0xDEADBEEFCAFEF00D)]
    with open(os.path.join(HERE, "field_constants.json"), "w") as f:
        json.dump(fld, f, indent=1)


if __name__ == "__main__":
    main()
