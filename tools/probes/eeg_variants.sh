#!/bin/bash
# tools/eeg_kernel_times.py under every tools/probes/build/libeav_*.so given (default: all) - run on the GPU box
cd "$(dirname "$0")/../.."
for so in ${@:-tools/probes/build/libeav_*.so}; do
  echo "== $so"
  EAV_LIB_PATH=$PWD/$so python3 tools/eeg_kernel_times.py 2>&1 | grep -v amdgpu.ids | head -${HEADN:-12}
  EAV_LIB_PATH=$PWD/$so python3 tools/eeg_kernel_times.py 2>&1 | grep "graph-replayed"
done
