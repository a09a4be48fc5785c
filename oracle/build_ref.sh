#!/bin/bash
# Builds oracle/_ref/librans64_ref.so from the reference's own vendored header, where it lies (no sources are copied).
# Runs only where /root/reference exists (the build container); the GPU box uses the prebuilt .so.
set -e
cd "$(dirname "$0")"
REF=${ICM_REFERENCE:-/root/reference}
if [ ! -f "$REF/third_party/ryg_rans/rans64.h" ]; then
  echo "oracle/_ref: reference header not present, keeping any prebuilt library"
  exit 0
fi
mkdir -p _ref
gcc -O2 -fPIC -shared -I "$REF/third_party/ryg_rans" rans64_shim.c -o _ref/librans64_ref.so
echo "built oracle/_ref/librans64_ref.so"
