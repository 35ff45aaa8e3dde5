#!/bin/bash
# A/B libraries of the specialised kernel's geometry for tools/ab_bench.py (tools/prebuilt/, git-ignored, shipped by gpurun):
#   tools/build_variants.sh name:"-DVPT_FAST_CC=1024 -DVPT_FAST_WG=7" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/prebuilt
SRC="model.cpp tables.cpp capi.cpp capi_device.cpp capi_host.cpp kernels.hip kernels_fast.hip kernels_tags.hip kernels_emit.hip"
for V in "$@"; do
  NAME="${V%%:*}"; DEFS="${V#*:}"
  ( cd vaporetto_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function $DEFS -o ../../tools/prebuilt/libvaporetto_$NAME.so $SRC ) &
done
wait
ls -la tools/prebuilt/
