#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -- builds sve_ntt_amd/build/lib_NAME.so from the
# current sources with extra flags on kernels.hip (A/B runs: SVENTT_HIP_LIBRARY=... python tools/quick_bench.py).
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
B=sve_ntt_amd/build; mkdir -p $B
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc"
/opt/rocm/bin/hipcc $F "$@" -c sve_ntt_amd/csrc/kernels.hip -o $B/kernels_$NAME.o &
for f in plan kernels_gold kernels_shoup; do
  [ -f $B/$f.o ] || /opt/rocm/bin/hipcc $F -c sve_ntt_amd/csrc/$f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/lib_$NAME.so $B/kernels_$NAME.o $B/plan.o $B/kernels_gold.o $B/kernels_shoup.o
echo $B/lib_$NAME.so
