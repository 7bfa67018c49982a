"""Perf contract of the hot kernels, checked WITHOUT a GPU.

The tile kernels run at 123-127 of 128 VGPRs on ~80 fixed-register asm constraints per statement
(csrc/gen_stage_asm.py); a compiler bump or an innocent edit can reintroduce register spills or a
longer instruction stream, and parity tests stay green while only speed dies (the FixedPoint64
kernels spilled 30-64 VGPRs for most of r02).  This test compiles the three kernel translation
units to gfx950 assembly (hipcc -S --cuda-device-only, no GPU needed) and asserts, for the hot
kernels of all three arithmetic back ends, against the shipped numbers in
tests/golden/perf_contract.json:

    .vgpr_spill_count <= shipped (0 for Montgomery and Goldilocks)
    .vgpr_count       <= 128     (four waves per SIMD: two 512-thread workgroups per CU)
    static VALU count <= shipped + 2 %   (straight-line kernels: static count == SQ_INSTS_VALU per wave)

    python tests/test_perf_contract.py --update     rewrites the fixture from the current build
"""
import json
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sve_ntt_amd", "csrc")
FIXTURE = os.path.join(ROOT, "tests", "golden", "perf_contract.json")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-S", "--cuda-device-only"]
UNITS = {"mont": "kernels.hip", "gold": "kernels_gold.hip", "shoup": "kernels_shoup.hip"}
ARITH = {"mont": 0, "gold": 1, "shoup": 2}

# (name, LOGT, F0, LOGL, MODE, FLAG): the kernels of BASELINE configs #3 (2^24 = col 2^11 x T4 | row 2^13,
# both directions), #4 (rows of 2^12) and #5's row phase (col 2^6 two-level)
HOT = [
    ("row 2^13 forward", 13, 0, 13, 0, 0, 0),
    ("row 2^13 inverse", 13, 0, 13, 1, 0, 0),
    ("col 2^11 x T4 forward", 13, 2, 11, 0, 1, 0),
    ("col 2^11 x T4 inverse", 13, 2, 11, 1, 1, 0),
    ("row 2^12 forward", 12, 0, 12, 0, 0, 0),
    ("row 2^12 inverse", 12, 0, 12, 1, 1, 0),
    ("col 2^6 x T64 two-level forward", 12, 6, 6, 0, 1, 1),
]


def _compile(unit: str, outdir: str) -> str:
    out = os.path.join(outdir, unit.replace(".hip", ".s"))
    subprocess.run([HIPCC, *FLAGS, os.path.join(CSRC, unit), "-o", out], check=True, capture_output=True)
    with open(out) as f:
        return f.read()


def _measure() -> dict:
    from sve_ntt_amd import build
    build.regenerate_stage_asm()
    with tempfile.TemporaryDirectory() as d, ThreadPoolExecutor(3) as ex:
        texts = dict(zip(UNITS, ex.map(lambda u: _compile(u, d), UNITS.values())))
    result = {}
    for backend, text in texts.items():
        # per-kernel metadata (the YAML note at the end of the file)
        meta = {}
        for m in re.finditer(r"\.name:\s+(_ZN10sventt_hip11tile_kernel\S+)(.*?)(?=\n  - \.a|\n\.\.\.|\Z)", text, re.S):
            block = m.group(2)
            spill = re.search(r"\.vgpr_spill_count:\s+(\d+)", block)
            vg = re.search(r"\.vgpr_count:\s+(\d+)", block)
            if spill and vg:
                meta[m.group(1)] = (int(vg.group(1)), int(spill.group(1)))
        bodies = {}
        for f in re.split(r"\n(?=_ZN10sventt_hip11tile_kernel[^\n]*:\s)", text):
            if f.startswith("_ZN"):
                bodies[f.split(":")[0]] = f.split("s_endpgm")[0]
        for name, logt, f0, logl, mode, flag, two in HOT:
            pat = (f"TileNTTILi{logt}ELi{f0}ELi{logl}ELi4ELi{mode}ELb{flag}ENS_5StepsIJ[^J]*EEELi{ARITH[backend]}"
                   f"ELb{two}EEE")
            hits = [k for k in meta if re.search(pat, k)]
            assert len(hits) == 1, (backend, name, hits)
            body = bodies[hits[0]]
            valu = sum(1 for ln in body.split("\n") if re.match(r"\s+v_", ln))
            vg, spill = meta[hits[0]]
            result[f"{backend}: {name}"] = {"vgpr": vg, "vgpr_spill": spill, "valu_static": valu}
    return result


@pytest.fixture(scope="module")
def measured():
    return _measure()


def test_hot_kernels_keep_their_registers_and_instruction_counts(measured):
    with open(FIXTURE) as f:
        shipped = json.load(f)["kernels"]
    assert set(shipped) == set(measured)
    bad = []
    for k, want in shipped.items():
        got = measured[k]
        if got["vgpr_spill"] > want["vgpr_spill"]:
            bad.append(f"{k}: {got['vgpr_spill']} spilled VGPRs (shipped {want['vgpr_spill']})")
        if got["vgpr"] > 128:
            bad.append(f"{k}: {got['vgpr']} VGPRs > 128 (four waves per SIMD lost)")
        if got["valu_static"] > want["valu_static"] * 1.02:
            bad.append(f"{k}: {got['valu_static']} static VALU > shipped {want['valu_static']} + 2 %")
    assert not bad, "\n".join(bad)


def test_montgomery_and_goldilocks_hot_kernels_do_not_spill(measured):
    """(r02's forward row 2^13 kernel parked one VGPR in scratch; with the <4,4,3,2> steps nothing does)"""
    for k, v in measured.items():
        if not k.startswith("shoup"):
            assert v["vgpr_spill"] == 0, (k, v)


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    m = _measure()
    for k, v in m.items():
        print(f"{k:48s} vgpr {v['vgpr']:3d}  spill {v['vgpr_spill']:2d}  VALU {v['valu_static']}")
    if "--update" in sys.argv:
        with open(FIXTURE, "w") as f:
            json.dump({"_doc": "Shipped register and static VALU counts of the hot tile kernels (gfx950, hipcc of the "
                               "ROCm 7.2 image); tests/test_perf_contract.py compares a fresh hipcc -S against these. "
                               "Regenerate with: python tests/test_perf_contract.py --update",
                       "kernels": m}, f, indent=1)
        print("wrote", FIXTURE)
