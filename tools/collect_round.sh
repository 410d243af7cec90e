#!/bin/bash
# Run on the GPU box (via gpurun): everything profiles/<tag>_* is made from, except the rocprofv3 passes
# (tools/profile.sh, tools/profile_routes.sh, tools/profile_mid_n.sh).  usage: tools/collect_round.sh r05 [full]
# "full" adds the frame-loop / graph-policy probes of round 2 (frozen since: the GUI they serve is out of scope).
set -u
TAG=${1:-r06}
FULL=${2:-}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q --timeout 900 --timeout-method=thread > $O/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
NB_HIP_FORCE_SHARDED=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 1 --steps 10 --warmup 2 > $O/${TAG}_rehearsal_1rank.json 2> $O/${TAG}_rehearsal_1rank.err; echo "rehearsal (socket rendezvous, /opt/rocm runtime) rc=$?"
NB_HIP_FORCE_SHARDED=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 \
    bench.py --gpus 1 --steps 10 --warmup 2 --rendezvous gloo > $O/${TAG}_rehearsal_1rank_gloo.json 2> $O/${TAG}_rehearsal_1rank_gloo.err; echo "rehearsal (gloo: torch's runtime first) rc=$?"
{
  echo "== nbody_amd/lib/nbody-bench (reference harness defaults: srand(11037), 10 warm-up + 100 steps, dt = 1; us/step), MI355X box, OMP_NUM_THREADS=16 =="
  OMP_NUM_THREADS=16 ./nbody_amd/lib/nbody-bench
  echo
  echo "== the same, GPU column only, fastest of 5 timed 100-step calls per world (--repeats 5) =="
  ./nbody_amd/lib/nbody-bench --gpu --repeats 5
  echo
  echo "== oracle/_ref/nbody-bench-ref: the reference's own src/bench.c + src/lib/world.c + sim_cpu.c + galaxy.c, linked against libnbody_hip.so =="
  OMP_NUM_THREADS=16 ./oracle/_ref/nbody-bench-ref
} > $O/${TAG}_nbody_bench_tables.txt 2>&1; echo "tables rc=$?"
python tools/fused_probe.py time 200 250 300 512 > $O/${TAG}_fused_chain.txt 2>&1; echo "fused rc=$?"
for P in 2 3; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 --master-port 2954$P bench.py --gpus $P \
      --transport host --steps 10 --warmup 2 > $O/${TAG}_rehearsal_${P}ranks_host.json 2> $O/${TAG}_rehearsal_${P}ranks_host.err; echo "host-transport P=$P rc=$?"
done
# round 5: bench.py started BARE (no launcher): a GPU-free supervisor starts the rank processes itself; with both ranks on this
# one GPU RCCL refuses the duplicate device, both ranks abort, and --transport auto answers with a fresh set of processes over
# the direct exchange (the line carries transport_fallback); then three ranks over the direct exchange asked for outright
python bench.py --gpus 2 --steps 10 --warmup 2 --extra-particles 262144 > $O/${TAG}_bare_2ranks_auto.json 2> $O/${TAG}_bare_2ranks_auto.err; echo "bare --gpus 2 (auto: rccl refused -> direct) rc=$?"
python bench.py --gpus 3 --transport direct --steps 10 --warmup 2 --extra-particles 262144 > $O/${TAG}_bare_3ranks_direct.json 2> $O/${TAG}_bare_3ranks_direct.err; echo "bare --gpus 3 --transport direct rc=$?"
# round 4: the C harness' own multi-process mode (ranks forked before any HIP call, shared page, no Python / torch)
{
  echo "== nbody-bench --gpus 1 --force-sharded --n 20000 --n 1048576 --steps 10 --warmup 2 --dt 0.01: the RCCL path with one rank (ncclCommInitRank, in-place ncclAllGather per step, overlapped step, chain captured as a hipGraph) on /opt/rocm's HIP runtime + librccl =="
  ./nbody_amd/lib/nbody-bench --gpus 1 --force-sharded --n 20000 --n 1048576 --steps 10 --warmup 2 --dt 0.01
  for P in 2 3; do
    echo
    echo "== nbody-bench --gpus $P --transport shm --n 4096 --n 65536 --n 1048576 --steps 10 --warmup 2 --dt 0.01: $P real C processes on ONE MI355X (host-staged exchange over the shared page; all ranks share the GPU, so the rate is one GPU's) =="
    ./nbody_amd/lib/nbody-bench --gpus $P --transport shm --n 4096 --n 65536 --n 1048576 --steps 10 --warmup 2 --dt 0.01
    echo
    echo "== nbody-bench --gpus $P --transport ipc (same sizes): the direct exchange -- every rank pushes its slice device-to-device into its peers' IPC-mapped source arrays, one barrier per step at the page =="
    ./nbody_amd/lib/nbody-bench --gpus $P --transport ipc --n 4096 --n 65536 --n 1048576 --steps 10 --warmup 2 --dt 0.01
  done
} > $O/${TAG}_nbody_bench_ranks.txt 2>&1; echo "C ranks rc=$?"
python tools/gpu_vs_avx.py > $O/${TAG}_gpu_vs_avx.txt 2>&1; echo "gpu-vs-avx rc=$?"
python -m pytest tests/test_gpu_zz_perf.py -q -s -m gpu -k near_the_best > $O/${TAG}_auto_vs_neighbours.txt 2>&1; echo "auto-vs-neighbours rc=$?"
[ "$FULL" = "full" ] || exit 0
{
  echo "== tools/frame_probe.py: the reference GUI's frame loop through include/nbody.h (300 frames each), defaults =="
  python tools/frame_probe.py 6000 1000 100000
  echo "== the same with round 1's behaviour: lazy read-back, timing events on every call, DMA upload (knobs set through nb_hip_tune and read back) =="
  NB_FRAME_KNOBS="readback=0,timing=1,zero_copy_upload=0" python tools/frame_probe.py 6000 1000 100000
  echo "== one knob at a time, N = 6000 =="
  NB_FRAME_KNOBS="readback=0" python tools/frame_probe.py 6000
  NB_FRAME_KNOBS="timing=1" python tools/frame_probe.py 6000
  NB_FRAME_KNOBS="zero_copy_upload=0" python tools/frame_probe.py 6000
  # (the wait policy NB_HIP_WAIT=spin|yield|block is read before the HIP context exists and only by TUNING=1 builds: not probed here)
  echo "== hipGraph chains in a frame loop: graph policy 0 (never) / 1 (always) / 2 (auto: 16+ steps, from the second use) =="
  for g in 0 1 2; do NB_FRAME_UPDATES=2,8,16,32 NB_HIP_GRAPH=$g NB_FRAME_KNOBS="readback=0" python tools/frame_probe.py 6000 | sed "s/^/[graph knob $g] /"; done
} > $O/${TAG}_frame_loop_latency.txt 2>&1; echo "frame rc=$?"
python tools/odd_chain_probe.py > $O/${TAG}_odd_chain_probe.txt 2>&1; echo "odd rc=$?"
python tools/first_call_probe.py > $O/${TAG}_first_call_probe.txt 2>&1; echo "first-call rc=$?"
python tools/graph_chunk_probe.py > $O/${TAG}_graph_chunk_probe.txt 2>&1; echo "chunk rc=$?"
{
  echo "== clock ramp-up after host-side world setup: nbody-bench --gpu --n 20000 --n 50000 with longer warm-up calls (reference: 10 steps, bench.c:21) =="
  for w in 10 400 1600; do ./nbody_amd/lib/nbody-bench --gpu --n 20000 --n 50000 --warmup $w | tail -2 | cut -f2,3,6 | tr "\n" " "; echo " <- warm-up steps: $w"; done
} > $O/${TAG}_warmup_ramp.txt 2>&1; echo "ramp rc=$?"
