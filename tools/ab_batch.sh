#!/bin/bash
# tools/ab_batch.sh OUT ROUNDS "quick_bench args" lib1 lib2 ... -- like ab.sh for another transform shape
OUT=$1; ROUNDS=$2; ARGS=$3; shift 3
rm -f $OUT
for r in $(seq $ROUNDS); do
  for L in "$@"; do
    echo "== $L" >> $OUT
    SVENTT_HIP_LIBRARY=$L timeout -k 10 200 python tools/quick_bench.py $ARGS >> $OUT 2>&1 || exit 1
  done
done
python - $OUT <<'PY'
import re, sys, collections, statistics
d = collections.defaultdict(lambda: collections.defaultdict(list))
lib = None
for line in open(sys.argv[1]):
    if line.startswith("== "):
        lib = line[3:].strip()
    if "MISMATCH" in line:
        print("MISMATCH in", lib)
    m = re.match(r"(forward|inverse) per-pass us: \[([^\]]*)\]", line)
    if m:
        d[lib][m.group(1)].append([float(x) for x in m.group(2).split(",")])
for lib, v in d.items():
    for k, runs in v.items():
        cols = list(zip(*runs))
        print("%-44s %s  min %s = %.1f   median %s = %.1f" % (
            lib, k, [min(c) for c in cols], sum(min(c) for c in cols),
            [round(statistics.median(c), 1) for c in cols], sum(statistics.median(c) for c in cols)))
PY
