#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh r02'): rocprofv3 kernel trace + the separate PMC passes the
# MI355X guide prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass).  Outputs under gpurun_out/prof_<tag>/; summarise
# afterwards with tools/summarise_profiles.py into profiles/.
TAG=${1:-r02}
ONLY=${2:-all}          # "eeg": the EEGNet passes only; "enc": the encoder passes only
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EEG="python3 $R/tools/eeg_steps.py 20"          # 3 eager + 20 graph-replayed train steps, nothing else
if [ "$ONLY" != "enc" ]; then
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eeg_trace -o eeg -- $EEG > $OUT/eeg_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/eeg_fetch -o eeg -- $EEG > $OUT/eeg_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/eeg_write -o eeg -- $EEG > $OUT/eeg_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/eeg_mfma -o eeg -- $EEG > $OUT/eeg_mfma.log 2>&1
fi
for K in "ast 8" "vit 128"; do
  [ "$ONLY" = "eeg" ] && break
  N=${K%% *}
  ENC="python3 $R/tools/encoder_step_bench.py $K split"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${N}_trace -o $N -- $ENC > $OUT/${N}_trace.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/${N}_fetch -o $N -- $ENC > $OUT/${N}_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/${N}_write -o $N -- $ENC > $OUT/${N}_write.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/${N}_mfma -o $N -- $ENC > $OUT/${N}_mfma.log 2>&1
  # the same step on ONE stream (NO_OVERLAP=1): kernel durations without the stretch two concurrent streams put on them
  NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${N}_serial -o $N -- $ENC > $OUT/${N}_serial.log 2>&1
done
# keep the merge-back small: counter / stats csv only
find $OUT -name "*.csv" -size +40M -delete
ls -la $OUT/*/ | head -60
