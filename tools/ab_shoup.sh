# A/B of libraries on the FixedPoint64 back end (62-bit test prime), N = 2^24: tools/bench_arith.py lines
for r in 1 2 3; do for L in "$@"; do echo "== $L"; SVENTT_HIP_LIBRARY=$L python tools/bench_arith.py 2>&1 | grep "fixed_point  2^24"; done; done
