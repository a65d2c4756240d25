#!/bin/bash
# time tools/fir_fft_bench.py under every tools/probes/build/libeav_*.so given (default: all) - run on the GPU box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for so in ${@:-tools/probes/build/libeav_*.so}; do
  echo "== $so"
  EAV_LIB_PATH=$PWD/$so python3 tools/fir_fft_bench.py 2>&1 | tail -4
done
