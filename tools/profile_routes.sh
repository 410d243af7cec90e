#!/bin/bash
# Run on the GPU box (via gpurun): the same rocprofv3 PMC passes over BOTH source routes of the step kernel
# (NB_HIP_VARIANT=1 scalar cache, =0 LDS tiles) at N = 2^20, so the routes' per-launch difference can be read from
# counters taken the same way on the same box.  Outputs under gpurun_out/prof_routes/; summarise afterwards with
# tools/summarize_routes.py <tag>.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_routes
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export NB_HIP_VARIANT=$v
  BENCH="python3 $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v${v}_stats -- $BENCH > $OUT/v${v}_stats.log 2>&1 || { echo "stats pass v=$v failed"; tail -5 $OUT/v${v}_stats.log; exit 1; }
  for pass in \
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU" \
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
      "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
      "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/v${v}_pmc_$tag -- $BENCH > $OUT/v${v}_pmc_$tag.log 2>&1 || { echo "pmc pass v=$v $pass failed"; tail -3 $OUT/v${v}_pmc_$tag.log; }
  done
done
find $OUT -name "*counter_collection.csv" | wc -l
