#!/bin/bash
# Run on the GPU box (via gpurun): the C harness' multi-process paths over ragged sizes, rank counts and transports.
# One wave per workgroup (--one-wave; the NB_HIP_W / NB_HIP_K presets exist in TUNING=1 builds only since ABI 0.3.0) makes the in-stream step independent of the launch geometry, so every
# "plain" verification must be bit-equal to the single-GPU World; "overlap" rows must stay within the harness' 1e-5.
# usage: tools/fuzz_ranks.sh > gpurun_out/r05_fuzz_ranks.txt
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
B=$R/nbody_amd/lib/nbody-bench
fail=0; runs=0
for T in shm ipc; do
  for P in 2 3 4; do
    for N in 200 333 777 1000 2111 4097 10000 30011; do
      runs=$((runs + 1))
      out=$(timeout -k 10 120 $B --gpus $P --transport $T --one-wave --n $N --steps 7 --warmup 2 --dt 0.01 --modes plain,overlap --verify 5 --own-rng --seed $((N + P)) 2>&1)
      rc=$?
      plain=$(echo "$out" | grep "mode=plain" | grep -c "ranks agree yes.*bitwise yes")
      over=$(echo "$out" | grep "mode=overlap" | grep -c "ranks agree yes")
      if [ $rc -ne 0 ] || [ "$plain" != "1" ] || [ "$over" != "1" ]; then
        fail=$((fail + 1)); echo "FAIL transport=$T P=$P N=$N rc=$rc"; echo "$out" | tail -8
      else
        echo "ok   transport=$T P=$P N=$N  $(echo "$out" | grep "mode=overlap" | sed 's/.*rel_l2_pos/overlap rel_l2_pos/')"
      fi
    done
  done
done
echo "fuzz_ranks: $runs runs, $fail failures"
[ $fail -eq 0 ]
