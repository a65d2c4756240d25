#!/bin/bash
# Build a full libeav_hip variant with ONE source recompiled under extra flags: tools/probes/build/libeav_<name>.so.
# usage: build_variant.sh <name> <source stem, e.g. eegnet_fir_fft> [extra hipcc flags ...]    (run HERE; `make` first)
# Load it on the GPU box with EAV_LIB_PATH=tools/probes/build/libeav_<name>.so (tools/ only, never the package).
set -e
cd "$(dirname "$0")/../.."
name=$1; stem=$2; shift 2
mkdir -p tools/probes/build
obj=tools/probes/build/${stem}_${name}.o
extra=""
[ "$stem" = attention_sp ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra "$@" \
  -c eav_amd/csrc/$stem.hip -o $obj
others=$(ls eav_amd/csrc/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $obj -o tools/probes/build/libeav_$name.so
echo built tools/probes/build/libeav_$name.so
