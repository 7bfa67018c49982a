#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -- builds sve_ntt_amd/build/lib_NAME.so from the
# current sources with extra flags on kernels.hip (A/B runs: SVENTT_HIP_LIBRARY=... python tools/quick_bench.py).
# The other three objects are shared between variants and rebuilt whenever any source under csrc/ or
# include/sventt_hip.h is newer than they are (a variant never mixes old and new objects).
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
B=sve_ntt_amd/build; mkdir -p $B
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc"
python - <<'PY'
import os, subprocess, sys
sys.path.insert(0, ".")
from sve_ntt_amd import build
build.regenerate_stage_asm()
PY
NEWEST=$(ls -t sve_ntt_amd/csrc/*.h sve_ntt_amd/csrc/*.inc sve_ntt_amd/csrc/*.hip include/sventt_hip.h | head -1)
/opt/rocm/bin/hipcc $F "$@" -c sve_ntt_amd/csrc/kernels.hip -o $B/kernels_$NAME.o &
for f in plan kernels_gold kernels_shoup; do
  if [ ! -f $B/$f.o ] || [ "$NEWEST" -nt $B/$f.o ]; then
    /opt/rocm/bin/hipcc $F -c sve_ntt_amd/csrc/$f.hip -o $B/$f.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/lib_$NAME.so $B/kernels_$NAME.o $B/plan.o $B/kernels_gold.o $B/kernels_shoup.o
echo $B/lib_$NAME.so
