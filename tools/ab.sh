#!/bin/bash
# tools/ab.sh OUT ROUNDS lib1 lib2 ... -- alternates tools/quick_bench.py 24 over the libraries (one process each,
# ROUNDS times) and prints, per library, the minimum and median per-pass times.  Run on the GPU box.
OUT=$1; ROUNDS=$2; shift 2
rm -f $OUT
for r in $(seq $ROUNDS); do
  for L in "$@"; do
    echo "== $L" >> $OUT
    SVENTT_HIP_LIBRARY=$L timeout -k 10 120 python tools/quick_bench.py 24 >> $OUT 2>&1 || exit 1
  done
done
python - $OUT <<'PY'
import re, sys, collections, statistics
d = collections.defaultdict(lambda: collections.defaultdict(list))
lib = None
for line in open(sys.argv[1]):
    if line.startswith("== "):
        lib = line[3:].strip()
    m = re.match(r"(forward|inverse) per-pass us: \[([^\]]*)\]", line)
    if m:
        d[lib][m.group(1)].append([float(x) for x in m.group(2).split(",")])
for lib, v in d.items():
    for k, runs in v.items():
        cols = list(zip(*runs))
        print("%-40s %s  min %s = %.1f   median %s = %.1f" % (
            lib, k, [min(c) for c in cols], sum(min(c) for c in cols),
            [round(statistics.median(c), 1) for c in cols], sum(statistics.median(c) for c in cols)))
PY
