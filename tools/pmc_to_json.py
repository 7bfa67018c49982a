#!/usr/bin/env python3
"""Refresh profiles/valu.json and the counter part of profiles/traffic.json from the rocprofv3 PMC
passes of tools/pmc_run.sh.   usage: tools/pmc_to_json.py <pmc dir> <summary path quoted as source>"""
import collections, csv, glob, json, os, sys

root, source = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"13, 2, 11, 4, 0, true": "col 2^11 x T4 (stride 8192)", "13, 0, 13, 4, 0, false": "row 2^13 (tile 2^13)"}
mean = collections.defaultdict(dict)
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    acc = collections.defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            for key, nm in NAMES.items():
                if "TileNTT<" + key in r["Kernel_Name"]:
                    acc[(nm, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (nm, c), v in acc.items():
        mean[nm][c] = sum(v) / len(v)

threads = (1 << 24) // 16
valu = json.load(open(os.path.join(here, "profiles", "valu.json")))
valu["source"] = source
for nm, m in mean.items():
    per_thread = m["SQ_INSTS_VALU"] * 64 / threads
    valu["kernels"][nm] = {
        "SQ_INSTS_VALU_per_launch": m["SQ_INSTS_VALU"], "GRBM_GUI_ACTIVE": m["GRBM_GUI_ACTIVE"],
        "instructions_per_thread": per_thread, "instructions_per_element": per_thread / 16,
        "cycles_per_instruction_per_simd": m["GRBM_GUI_ACTIVE"] / 8 * 1024 / m["SQ_INSTS_VALU"],
        "issue_cost_of_the_mix_cycles": valu["kernels"].get(nm, {}).get("issue_cost_of_the_mix_cycles", 4.0)}
json.dump(valu, open(os.path.join(here, "profiles", "valu.json"), "w"), indent=2)

tr = json.load(open(os.path.join(here, "profiles", "traffic.json")))
tr["source"] = source
for nm, m in mean.items():
    c = tr["counters"].setdefault(nm, {"fetch_correction": 2})
    c.update({"FETCH_SIZE_KiB": m["FETCH_SIZE"], "WRITE_SIZE_KiB": m["WRITE_SIZE"],
              "TCC_EA0_RDREQ_sum": m["TCC_EA0_RDREQ_sum"], "TCC_EA0_WRREQ_sum": m["TCC_EA0_WRREQ_sum"]})
    tr["bytes_per_launch"][nm] = int(round((m["FETCH_SIZE"] * c["fetch_correction"] + m["WRITE_SIZE"]) * 1024))
json.dump(tr, open(os.path.join(here, "profiles", "traffic.json"), "w"), indent=2)
print(json.dumps({k: v["instructions_per_thread"] for k, v in valu["kernels"].items()}), tr["bytes_per_launch"])
