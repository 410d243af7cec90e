#!/bin/bash
# Run on the GPU box (via gpurun): the before / after counter pair of the persistent-launch experiment:
# kernel-trace + SQ counters of 110 plain-launch steps at N = 10 000 and 20 000, classic launch (persist 1) and persist 2.
# Outputs under gpurun_out/prof_persist/; summarise with tools/summarize_persist.py.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_persist
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 10000 20000; do
  for p in 1 2; do
    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv \
        -d $OUT/n${n}_p${p} -- python3 $R/tools/persist_pmc_driver.py $n $p > $OUT/n${n}_p${p}.log 2>&1 || { echo "pass n=$n p=$p failed"; tail -5 $OUT/n${n}_p${p}.log; }
  done
done
find $OUT -name "*counter_collection.csv" | head
