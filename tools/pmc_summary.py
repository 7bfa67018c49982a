#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            if "tile_kernel" not in name:
                continue
            short = name.split("TileNTT<")[1].split(">")[0]
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.relpath(f, root))
    for k, d in agg.items():
        print("  TileNTT<%s>" % k)
        for c, v in d.items():
            print("    %-24s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
