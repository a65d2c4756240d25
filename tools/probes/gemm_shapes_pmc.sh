#!/bin/bash
# ON THE GPU BOX: per-product counter passes (HBM reads / writes, L2 hits) over tools/probes/gemm_shapes.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/gemm_shapes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
KIND=${1:-vit}
BATCH=$2
i=0
DBS=""
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf $OUT/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o p -- python3 $R/tools/probes/gemm_shapes.py $KIND $BATCH > $OUT/p$i.log 2>&1
  DB=$(find $OUT/p$i -name "*.db" | head -1)
  [ -n "$DB" ] && DBS="$DBS $DB" || tail -3 $OUT/p$i.log
done
python3 $R/tools/probes/gemm_shapes_report.py $KIND $BATCH $DBS
find $OUT -name "*.db" -delete
