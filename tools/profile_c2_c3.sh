#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of the kernels BASELINE.json's configs 2 and 3
# actually run, through the C harness with every knob on auto (the program itself sits right after `--`):
#   C2  nbody-bench --gpu --n 65536  --dt 0.01                                   (10 warm-up + ONE 100-step call, bench.c:21-35)
#   C3  nbody-bench --gpu --n 262144 --dt 0.01 --steps 20 --warmup 20 --repeats 3 (20-step chain: plain, then a cached hipGraph)
#       + the same at --dt 0.005 (the dt-halved leg of config 3)
# --floor-rate is given so that no calibration world (N = 100 000) adds launches to the per-kernel rows.
# Outputs under gpurun_out/prof_c23/{c2,c3}/; summarise with tools/summarize_c2_c3.py <tag>.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_c23
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$R/nbody_amd/lib/nbody-bench
declare -A CMD
CMD[c2]="$B --gpu --n 65536 --dt 0.01 --floor-rate 5.5e12"
CMD[c3]="$B --gpu --n 262144 --dt 0.01 --steps 20 --warmup 20 --repeats 3 --floor-rate 5.5e12"
CMD[c3h]="$B --gpu --n 262144 --dt 0.005 --steps 20 --warmup 20 --repeats 3 --floor-rate 5.5e12"
for c in c2 c3 c3h; do
  mkdir -p $OUT/$c
  echo "${CMD[$c]}" | sed "s#$R/##" > $OUT/$c/command.txt
  ${CMD[$c]} > $OUT/$c/plain_run.txt 2>&1 || { echo "$c: unprofiled run failed"; exit 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c/stats -- ${CMD[$c]} > $OUT/$c/stats.log 2>&1 || { echo "$c: stats pass failed"; tail -5 $OUT/$c/stats.log; exit 1; }
  echo "$c stats done"
  [ $c = c3h ] && continue
  i=0
  for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$c/pmc_$i -- ${CMD[$c]} > $OUT/$c/pmc_$i.log 2>&1 || { echo "$c: pmc pass '$pass' failed"; tail -5 $OUT/$c/pmc_$i.log; }
    echo "$c pmc $i done"
  done
done
find $OUT -name "*.csv" | wc -l
