#!/bin/bash
# Build libicm_hip.so for gfx950 (MI355X). Cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
mkdir -p lib build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result"
objs=""
pids=""
for f in conv_igemm conv_1x1 conv_ks8 conv_wino conv_wgrad wgrad_wino pointwise entropy winattn winattn_mfma; do
  stale=0
  for dep in csrc/$f.hip csrc/icm_common.h ../include/icm_hip.h $(grep -q conv_common.h csrc/$f.hip && echo csrc/conv_common.h); do
    if [ ! -f build/$f.o ] || [ $dep -nt build/$f.o ]; then stale=1; fi
  done
  if [ $stale = 1 ]; then
    rm -f build/$f.o
    $HIPCC $FLAGS -c csrc/$f.hip -o build/$f.o &
    pids="$pids $!"
  fi
  objs="$objs build/$f.o"
done
if [ ! -f build/rans.o ] || [ csrc/rans.cpp -nt build/rans.o ] || [ ../include/icm_hip.h -nt build/rans.o ]; then
  g++ -O2 -fPIC -std=c++17 -Wall -c csrc/rans.cpp -o build/rans.o || { echo "compile failed"; exit 1; }
fi
objs="$objs build/rans.o"
for p in $pids; do wait $p || { echo "compile failed"; exit 1; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o lib/libicm_hip.so $objs
echo "built lib/libicm_hip.so"
