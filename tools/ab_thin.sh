L=sve_ntt_amd/build/lib_thin.so
for r in 1 2 3; do
echo "== base"; SVENTT_HIP_LIBRARY=$L python tools/quick_bench.py 24 2>&1 | grep -E "plan|digest|round|per-pass"
echo "== thin X"; SVENTT_COL_THIN=1 SVENTT_HIP_LIBRARY=$L python tools/quick_bench.py 24 2>&1 | grep -E "plan|digest|round|per-pass"
echo "== thin Y 12,12"; SVENTT_SPLIT=12,12 SVENTT_COL_THIN=1 SVENTT_HIP_LIBRARY=$L python tools/quick_bench.py 24 2>&1 | grep -E "plan|digest|round|per-pass"
echo "== slim 12,12"; SVENTT_SPLIT=12,12 SVENTT_HIP_LIBRARY=$L python tools/quick_bench.py 24 2>&1 | grep -E "plan|digest|round|per-pass"
done
