#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + one PMC pass of nbody-bench at the mid sizes of the
# reference's table (N = 10 000, 20 000; bench.c:38), to back the latency-bound model of choose_shape's small-launch
# branch.  Outputs under gpurun_out/prof_mid/; summarise with tools/summarize_mid_n.py.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_mid
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# SIZES="--n 500 --n 2000 --n 4000" tools/profile_mid_n.sh profiles other rows (the lane-split launches of round 3)
SIZES=${SIZES:-"--n 10000 --n 20000 --n 50000"}
BENCH="$R/nbody_amd/lib/nbody-bench --gpu $SIZES --steps 100 --warmup 10 --dt 0.01 --floor-rate 5.5e12"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/stats.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || { echo "pmc pass failed"; tail -5 $OUT/pmc_sq.log; }
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/pmc_grbm -- $BENCH > $OUT/pmc_grbm.log 2>&1 || { echo "grbm pass failed"; tail -5 $OUT/pmc_grbm.log; }
find $OUT -name "*.csv" | head -20
