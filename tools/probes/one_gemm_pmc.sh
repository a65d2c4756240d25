#!/bin/bash
# ON THE GPU BOX: counter passes over one product (tools/probes/one_gemm.py); per-kernel means by tools/rocpd_pmc.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/one_gemm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$@"
i=0
if [ -n "$QUICK" ]; then set -- "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; else set -- "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum"; fi
for PMC in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o p -- python3 $R/tools/probes/one_gemm.py $ARGS > $OUT/p$i.log 2>&1
  DB=$(find $OUT/p$i -name "*.db" | head -1)
  echo "== $PMC"
  if [ -n "$DB" ]; then python3 $R/tools/rocpd_pmc.py $DB gemm_sp | grep -v "^    launches" ; else tail -3 $OUT/p$i.log; fi
done
find $OUT -name "*.db" -delete
