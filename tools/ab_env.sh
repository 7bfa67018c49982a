B=sve_ntt_amd/build
for r in 1 2 3; do for t in 10 8 9 11 12; do echo "== twist_lo $t"; SVENTT_TWIST_LO_LOG2=$t SVENTT_HIP_LIBRARY=$B/lib_base.so python tools/quick_bench.py 24 2>&1 | grep per-pass; done; done
