#!/bin/bash
# ON THE GPU BOX: SQ counters of the FFT FIR kernels (tools/fir_fft_bench.py); per-kernel means by tools/rocpd_pmc.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/fir_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  rm -rf $OUT/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o p -- python3 $R/tools/fir_fft_bench.py > $OUT/p$i.log 2>&1
  DB=$(find $OUT/p$i -name "*.db" | head -1)
  echo "== $PMC"
  if [ -n "$DB" ]; then python3 $R/tools/rocpd_pmc.py $DB fir_fft_ | grep -v "^    launches" ; else tail -3 $OUT/p$i.log; fi
done
find $OUT -name "*.db" -delete
