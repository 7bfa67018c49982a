#!/usr/bin/env python3
"""LDS bank-conflict degree of every LDS exchange of a tile shape under a swizzle.
For each step that touches LDS: the ds_*_b64 of element v of set g by the 32 lanes of a
half-wave; degree = max lanes on one of the 32 8-byte bank pairs (1 = conflict free)."""

def phys_default(I):
    return I ^ ((I >> 4) & 31)

def analyse(logt, f0, logl, loge, steps, phys=phys_default):
    nt = 1 << (logt - loge)
    E = 1 << loge
    out = []
    hi_rel = logl
    for si, k in enumerate(steps):
        lo_rel = hi_rel - k
        lo, hi = f0 + lo_rel, f0 + hi_rel
        worst, total, cnt = 0, 0, 0
        for v in range(1 << k):
            for g in range(E >> k):
                for wave in range(0, nt, 64):
                    for half in (0, 32):
                        if wave + half >= nt:
                            continue
                        banks = {}
                        for lane in range(min(32, nt - wave - half)):
                            s = wave + half + lane + g * nt
                            I = ((s >> lo) << hi) | (v << lo) | (s & ((1 << lo) - 1))
                            b = phys(I) & 31
                            banks[b] = banks.get(b, 0) + 1
                        m = max(banks.values())
                        worst = max(worst, m); total += m; cnt += 1
        out.append((k, worst, total / cnt))
        hi_rel = lo_rel
    return out

if __name__ == "__main__":
    shapes = {
        "row 2^13 (4,4,3,2)": (13, 0, 13, 4, (4, 4, 3, 2)),
        "row 2^13 (4,4,4,1) r02": (13, 0, 13, 4, (4, 4, 4, 1)),
        "row 2^12 (4,3,3,2)": (12, 0, 12, 4, (4, 3, 3, 2)),
        "col 2^11 T4 (4,3,4)": (13, 2, 11, 4, (4, 3, 4)),
        "col 2^11 T4 (4,4,3) r02": (13, 2, 11, 4, (4, 4, 3)),
        "col 2^11 T8 (4,4,3)": (14, 3, 11, 4, (4, 4, 3)),
        "col 2^10 T4 (4,4,2)": (12, 2, 10, 4, (4, 4, 2)),
        "col 2^9 T8 (4,4,1)": (12, 3, 9, 4, (4, 4, 1)),
        "col 2^8 T16 (4,4)": (12, 4, 8, 4, (4, 4)),
        "col 2^7 T32 (4,3)": (12, 5, 7, 4, (4, 3)),
        "col 2^6 T64 (4,2)": (12, 6, 6, 4, (4, 2)),
        "col 2^5 T128 (4,1)": (12, 7, 5, 4, (4, 1)),
        "row 2^8 in 2^12 (4,4)": (12, 0, 8, 4, (4, 4)),
        "row 2^9 in 2^12 (4,4,1)": (12, 0, 9, 4, (4, 4, 1)),
        "row 2^10 in 2^12 (4,4,2)": (12, 0, 10, 4, (4, 4, 2)),
        "row 2^11 in 2^12 (4,4,3)": (12, 0, 11, 4, (4, 4, 3)),
        "row 2^7 in 2^12 (4,3)": (12, 0, 7, 4, (4, 3)),
    }
    for name, sh in shapes.items():
        r = analyse(*sh)
        # exchanges: write side of step i and read side of step i+1 (HBM on the outer sides)
        desc = []
        for i, (k, worst, avg) in enumerate(r):
            sides = []
            if i > 0: sides.append("rd")
            if i + 1 < len(r): sides.append("wr")
            if sides: desc.append(f"k={k}[{'/'.join(sides)}] worst {worst} avg {avg:.2f}")
        print(f"{name:28s} " + " | ".join(desc))
