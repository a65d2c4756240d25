#!/bin/bash
# encoder step time (tools/encoder_step_bench.py, split precision) under each given library - run on the GPU box
cd "$(dirname "$0")/../.."
SHAPES=${SHAPES:-"vit 128;ast 8"}
for so in "$@"; do
  IFS=';' read -ra SH <<< "$SHAPES"
  for sh in "${SH[@]}"; do
    echo -n "$(basename $so)  "
    EAV_LIB_PATH=$PWD/$so python3 tools/encoder_step_bench.py $sh split 2>&1 | tail -1
  done
done
