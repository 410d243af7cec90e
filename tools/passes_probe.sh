#!/bin/bash
# FETCH_SIZE and step time of bench.py for several source-pass counts (NB_HIP_PASSES: needs a `make TUNING=1` build since ABI 0.3.0); run on the GPU box.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/passes; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in 1 2 3 4 0; do
  export NB_HIP_PASSES=$P
  python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('passes=$P', 'ms_per_step=%.2f'%d['ms_per_step'], 'int/s=%.3e'%d['value'], 'launches', d['roofline']['launches'])"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p$P -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/p$P.log 2>&1
  python3 - <<PY
import csv, glob
rows=[r for f in glob.glob('$OUT/p$P/*/*_counter_collection.csv') for r in csv.DictReader(open(f)) if 'step_kernel' in r['Kernel_Name']]
tot=sum(float(r['Counter_Value']) for r in rows)
steps=4
print('  passes=$P FETCH_SIZE per STEP: %.1f MB raw over %d launches' % (tot*1024/1e6/steps, len(rows)))
PY
done
