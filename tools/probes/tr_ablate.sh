#!/bin/bash
# Build timing-only ablation variants of gemm_sp.hip (EAV_ABL bits: 1 no fragment reads, 2 no LDS-DMA, 4 no MFMAs,
# 8 no L2 prefetch) as stand-alone libraries under tools/probes/build/ - run HERE (hipcc), then tr_ablate.py on the GPU box.
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/probes/build
for abl in ${ABLS:-0 1 2 3 4 8 5 6}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -DEAV_ABL=$abl \
    eav_amd/csrc/gemm_sp.hip eav_amd/csrc/eav_common.hip -o tools/probes/build/libgemm_abl$abl.so &
done
wait
ls -la tools/probes/build
