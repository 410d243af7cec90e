#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
# Outputs under gpurun_out/prof/; summarise afterwards with tools/summarize_profile.py.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-clock-probe"   # no probe / sampler legs: the per-kernel rows hold the step kernels only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH --no-extra-configs --steps 10 --warmup 2 > $OUT/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/stats.log; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$tag -- $BENCH --no-extras --steps 3 --warmup 1 > $OUT/pmc_$tag.log 2>&1 || { echo "pmc pass $pass failed"; tail -5 $OUT/pmc_$tag.log; }
done
find $OUT -name "*.csv" | head -40
