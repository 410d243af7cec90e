#!/usr/bin/env python3
"""bench.py -- the hot path's headline number: particle-pair interactions/s at N = 2^20 on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one process per GPU
through torch.distributed.run.  One JSON line on rank 0.  `python bench.py --gpus N` started BARE works too: the process
that is started never touches the GPU -- it is a supervisor over fresh rank processes (nbody_amd/launch.py, DESIGN.md
section 4), which is also what every rank process is under torch.distributed.run.  The whole run lives inside ONE budget
(--budget-s, default 480 s: under the 600 s a driver allows the command) and always prints its line.

Workload (BASELINE.json metric): srand(11037); MakeGalaxies(2^20, 2) -- the reference bench's universe
(src/bench.c:42,53) at the size the metric is quoted on -- partitioned by CreateWorld, dt = 0.01.  A "step" is one force +
integrate pass over all N receivers against all mass_len sources.  Interactions per step = N * mass_len (what the
reference kernels evaluate, particle_cs.glsl:30,35).  K steps run as ONE PerformSimUpdate(K) call, like the reference
harness' update(w, dt, 100) (bench.c:30-33); particles are resident in HBM before the timed region (SetSimulationData is
outside it).

N = 1: the CPU baseline runs FIRST (the reference's own UpdateWorld_CPU, compiled where it lies, the product's on the
same World and thread count, and -- informational -- the fastest CPU variant the host runs: cpu_baseline.best_cpu), then the
GPU legs back to back: headline K steps -- from here on the line is in hand and
every later leg runs under a C-level last-gasp handler --, the clock probe, the headline's parity stamp, two repeats of
the same K steps (run-to-run spread), the same K steps with the clock sampler beside them (roofline.held_clock_ghz,
cycles_per_wave_interaction), the LDS-tile route (roofline.alt_lds), and `extra_configs`: BASELINE.json's other
single-GPU configurations in the reference harness' shape -- C2 (N = 65 536, one 100-step call after 10 warm-up steps),
C3 (N = 262 144, a cached hipGraph chain at dt, then the same chain at dt/2), N2 (all-massive), C1 (N = 4 096, the
product's nbody-bench --cpu, no GPU) -- and the per-rank shard steps S2 / S4 / S8 / C5S8: what ONE rank's step of a 2 / 4 /
8-way sharded run costs, measured on this one GPU (the compute half of the 1/2/4/8 curve).

N > 1: strong scaling -- the same 2^20 particles, N/P receivers per GPU, all-gather of source positions per step over RCCL
inside the library.  `--transport auto` (default) = rccl -> direct -> host: an attempt that does not deliver a verified
headline (a rank that leaves during bring-up, a time-out, a failed self-check) is followed by a FRESH set of rank processes
over the next transport; the line carries one "transport_fallback" entry per step, and a failed self-check stays on it as
"verification_failed" (exit code 5 whatever follows).  Before the first contact every rank writes its "preflight" record
(PCI address, peer-access row, one IPC open of the next rank's word, RCCL bring-up timings).  The rendezvous, the barriers and the reductions of the timings go over a stdlib Unix-socket hub
between the ranks (nbody_amd/ranklink.py): torch is NOT imported, so the HIP runtime and the librccl the data path binds
are /opt/rocm's -- the stack every single-GPU test runs on (`--rendezvous gloo` keeps the round-3 route).  The JSON dict
is COMPLETE after the headline leg and the self-check (mandatory for every N > 1 headline: all ranks agree and match a
single-GPU run; what the RCCL communicator itself reports; per-step kernel / all-gather times); every later leg
(`extra_configs`: overlapped step, the chain captured as a hipGraph, BASELINE.json's config 5 at N = 2^22 plain and
overlapped, the direct exchange with its own self-check) runs under a host-side deadline -- as does the headline leg itself,
inside the attempt's share of the budget: if one stalls, rank 0 writes the line with what is in hand plus "extras_aborted":
"<leg>" (or, with no headline yet, value null and why) and every rank leaves with exit code 4 -- a fresh exit, never a
re-exec.  Failure rehearsals are not in this file: tests/bench_rehearsal.py registers observers when NB_BENCH_REHEARSE is set.

`runtime` in the JSON line says which HIP runtime and librccl the run bound.

Every single-GPU leg carries a parity stamp: OUTSIDE the timed region the state the timed steps left is read back, one
more step runs, and that step is checked -- sampled accelerations against the oracle's float64 sum (the stated
tolerance) and against the reference AVX order, the integrator identity bit for bit over all N particles.

The oracle (oracle/) is used here ONLY as the timed cpu_baseline and as the checker behind the parity stamps; nothing
that is timed as "the GPU path" goes through it.
"""
import argparse
import ctypes as C
import datetime
import hashlib
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nbody_amd.benchlib import (FLOP_PER_INTERACTION, GATHER_LATENCY_ASSUMED_MS, PEAK_FP32_VECTOR_TFLOPS, XGMI_LINK_GBS,  # noqa: E402,F401
                                algorithmic_bytes_per_launch, cpu_model, device_cus, gather_estimate_ms, held_clock_fields,
                                host_cores, host_cpu_share, kernel_sources_sha, libgomp, pmc_traffic, pmc_traffic_parts)
from nbody_amd.launch import LastGasp, LegGuard, headline_of as _headline_of, supervise  # noqa: E402,F401

N_PARTICLES = 1 << 20
N_CONFIG5 = 1 << 22
N_CONFIG2 = 1 << 16
N_CONFIG3 = 1 << 18
DT = 0.01

# Announcements of what the run is about to do.  EMPTY in every real run; tests/bench_rehearsal.py (imported only when
# NB_BENCH_REHEARSE is set) registers observers that make a chosen failure happen at a chosen place.
OBSERVERS = []


def notify(event, **info):
    for observer in OBSERVERS:
        observer(event, info)


def make_workload(n, all_massive=False):
    """Product code only: MakeGalaxies + CreateWorld's partition (no GPU touched)."""
    import nbody_amd as nb

    ic = nb.make_galaxies(n, 2, seed=11037)
    if all_massive:
        # the N^2 run of the N-body literature.  The massless half gets the mass galaxy.h would
        # give a body of its radius (NP_R_TO_M(0.5) = 4*pi*10/3 * 0.125), so every particle is a source.
        light = ic[:, 6] <= 0
        ic[light, 6] = np.float32(4.0 * np.pi * 10.0 / 3.0) * ic[light, 7] ** 3
    w = nb.World(ic)
    part = w.particles()
    w.close()
    mass_len = int((part[:, 6] > 0).sum())
    return part, mass_len


# ---- CPU baseline ---------------------------------------------------------------------------------------------------

def cpu_baseline(part, mass_len, budget_s=12.0):
    """The reference's own UpdateWorld_CPU (src/lib/world.c:99-110: PackParticles + the OpenMP schedule(static, 20) loop
    over PackedUpdate, src/lib/sim_cpu.c:156-194), compiled where it lies into oracle/_ref/libnbody_ref_world.so, timed
    on this box's host cores over a bounded sample of the same workload; beside it the product's UpdateWorld_CPU on the
    same World with the same number of threads, and -- informational -- the fastest of the product's CPU variants this
    host runs (`best_cpu`).  When oracle/_ref is absent: the oracle's AVX restatement (kind "port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    share = host_cpu_share()
    cores = share["threads"]
    n = part.shape[0]
    # ~2e9 interactions/s/core: size the receiver sample for about budget_s seconds.  The sample is
    # a World of the first `recv` partitioned particles: massive particles come first, so it holds ALL mass_len sources
    # and one UpdateWorld_CPU step of it evaluates exactly recv x mass_len interactions.
    recv = int(budget_s * 2.0e9 * cores / max(mass_len, 1))
    recv = max(mass_len, min(n, recv // (20 * cores) * (20 * cores)))
    kind, threads, sec, one, how = "port", cores, None, None, None
    if os.path.exists(ob.REF_WORLD_SO):
        try:
            # untimed: bring the OpenMP team up and the host cores out of idle (on a virtualised host the first parallel
            # region after an idle spell can run as good as serialised for a second) -- small steps for ~0.6 s
            t_warm = time.perf_counter()
            while time.perf_counter() - t_warm < 0.6:
                _time_reference_world(ob.REF_WORLD_SO, part, min(n, 8192), cores, steps=20)
            sec = _time_reference_world(ob.REF_WORLD_SO, part, recv, cores)
            kind = "reference"
            how = ("reference UpdateWorld_CPU (src/lib/world.c:99-110 + sim_cpu.c, compiled where it lies: "
                   "oracle/_ref/libnbody_ref_world.so; its OpenMP loop, omp_set_num_threads(%d))" % cores)
            # SURVEY.md 8d also asks for the 1-thread figure: ~2 s of the same call on one thread
            recv1 = max(64, min(n, int(2.0 * 2.0e9 / max(mass_len, 1))))
            one = recv1 * min(mass_len, recv1) / _time_reference_world(ob.REF_WORLD_SO, part, recv1, 1)
        except Exception as e:  # pragma: no cover - diagnostic only
            print(f"[bench] reference UpdateWorld_CPU leg failed ({e}); falling back", file=sys.stderr)
            sec = None
    if sec is None:
        sec, threads, _ = ob.time_avx_sample(part, mass_len, 0, recv, dt=DT, threads=cores)
        how = "oracle/nbody_oracle.c AVX restatement (bit-exact with the reference's AVX build)"
    out = {
        "value": recv * mass_len / sec,
        "unit": "interactions/s",
        "cores": threads,
        "kind": kind,
        "how": how,
        "cpu_model": cpu_model(),
        "value_1_thread": one,
        "sample": f"World of the first {recv} of {n} partitioned particles (all {mass_len} sources), one step, AVX (-mavx, no FMA) "
                  f"+ {threads} threads ({sec:.2f} s); the whole step would take ~{sec * n / recv:.0f} s",
    }
    out.update({k: share[k] for k in ("threads_from", "affinity_cores", "os_cpu_count", "cgroup_cpu_quota", "OMP_NUM_THREADS")})
    for key, leg in (("product", product_cpu_leg), ("best_cpu", best_cpu_leg)):
        try:
            out[key] = leg(part, mass_len, recv, threads)
        except Exception as e:  # pragma: no cover - diagnostic only
            out[key] = {"error": str(e)}
    return out


def _time_reference_world(so, part, recv, threads, steps=1):
    """Seconds of ONE UpdateWorld_CPU(dt, steps) of the reference's own world.c on a World of part[:recv].  The library's five
    sim_gpu.h symbols bind to libnbody_hip.so (loaded first, RTLD_GLOBAL); CreateSimPipeline allocates nothing on a GPU and
    nothing else of the seam is called by a CPU step, so no GPU is touched.  RTLD_DEEPBIND: the reference's calls into its
    own sim_cpu.c must resolve inside its library, whatever else this process has loaded."""
    import nbody_amd as nb

    nb.hip_lib()
    ref = C.CDLL(so, mode=os.RTLD_NOW | os.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0))
    ref.CreateWorld.restype = C.c_void_p
    ref.CreateWorld.argtypes = [C.c_void_p, C.c_uint32]
    ref.UpdateWorld_CPU.restype = None
    ref.UpdateWorld_CPU.argtypes = [C.c_void_p, C.c_float, C.c_uint32]
    ref.DestroyWorld.restype = None
    ref.DestroyWorld.argtypes = [C.c_void_p]
    gomp = libgomp()
    if gomp is not None:
        gomp.omp_set_num_threads(C.c_int(threads))
    sample = np.ascontiguousarray(part[:recv])
    w = ref.CreateWorld(sample.ctypes.data, recv)   # copies and partitions (already partitioned: order unchanged)
    try:
        t0 = time.perf_counter()
        ref.UpdateWorld_CPU(w, DT, steps)
        return time.perf_counter() - t0
    finally:
        ref.DestroyWorld(w)


def product_cpu_leg(part, mass_len, recv, threads):
    """The product's own CPU path (libnbody.so: UpdateWorld_CPU = AVX + OpenMP, bit-exact with the reference's AVX
    build) on the same receiver sample: a World of the first `recv` partitioned particles holds ALL mass_len sources
    (massive particles come first) and steps exactly recv x mass_len interactions.  No GPU is touched."""
    import nbody_amd as nb

    recv = max(recv, mass_len)  # the sources must all be in the World
    gomp = libgomp()
    if gomp is not None:
        gomp.omp_set_num_threads(C.c_int(threads))
    w = nb.World(part[:recv])
    try:
        t0 = time.perf_counter()
        w.update_cpu(DT, 1)
        sec = time.perf_counter() - t0
    finally:
        w.close()
    return {"value": recv * mass_len / sec, "unit": "interactions/s", "cores": threads if gomp is not None else None,
            "kind": "product (UpdateWorld_CPU, nbody_amd/csrc/sim_cpu.c, AVX + OpenMP, bit-exact with the reference AVX build)",
            "sample": f"World of the first {recv} partitioned particles (all {mass_len} sources), one UpdateWorld_CPU step ({sec:.2f} s)"}


# ---- parity stamp ----------------------------------------------------------------------------------------------------

def best_cpu_leg(part, mass_len, recv, threads):
    """SURVEY.md 8d's informational "best CPU" row: the product's sim_cpu.c built for what the host cores can do (FMA
    contraction, 16 lanes, rsqrt estimate + Newton: nbody_amd/csrc/cpu_best.c; the reference's whole SIMD matrix is
    src/lib/CMakeLists.txt:24-33), every variant this CPU runs timed on the same one-step sample with the same threads.
    NOT bit-exact with any reference build -- within the stated fp32 tolerance (tests/test_world_cpu.py) -- and never the
    stated baseline: that stays the -mavx reference row."""
    import nbody_amd as nb

    recv = max(recv, mass_len)
    gomp = libgomp()
    if gomp is not None:
        gomp.omp_set_num_threads(C.c_int(threads))
    sample = np.ascontiguousarray(part[:recv])
    rates, skipped = {}, []
    for isa, runs_here, _what in nb.cpu_variants():
        if not runs_here:
            skipped.append(isa)
            continue
        t0 = time.perf_counter()
        nb.cpu_variant_update(isa, sample, mass_len, DT, 1)
        rates[isa] = recv * mass_len / (time.perf_counter() - t0)
    if not rates:
        return {"value": None, "note": "this CPU runs none of the variants", "not_supported_here": skipped}
    best = max(rates, key=rates.get)
    return {"value": rates[best], "unit": "interactions/s", "isa": best, "cores": threads, "variants": rates, "not_supported_here": skipped,
            "what": dict((i, w) for i, _, w in nb.cpu_variants())[best],
            "note": "not bit-exact with the reference; informational (within the stated fp32 tolerance of the float64 sum); "
                    "the stated baseline is the -mavx reference row above"}


def parity_stamp(sim, mass_len, dt=DT, samples=256):
    """What the line says about the correctness of what it timed.  Reads the state S the timed steps left, runs ONE more
    step on the same pipeline (same launch shape, same route) and checks that step against the oracle (the checker,
    tests/oracle_binding.py -- never the thing measured):
      * acc of `samples` receivers (half massive, half anywhere, plus the first / last of each range) against the
        float64 sum of the same state: worst |acc - acc_f64| / (1e-4 |acc_f64| + 1e-6 sum_j |contribution_j|), the stated
        one-step tolerance (<= 1 passes);
      * the same receivers against the reference AVX order's fp32 result, as a fraction of sum_j |contribution_j|;
      * vel == vel0 + acc * dt and pos == pos0 + vel * dt in fp32 with the reference's roundings (sim_cpu.c:191-193) and
        mass / radius untouched, over ALL particles, bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    before = sim.get_data()
    if hasattr(sim, "update"):
        sim.update(1, dt)
    else:
        sim.step(1, dt)    # a LocalShardGroup: all P shards of one world advance together
    after = sim.get_data()
    n = before.shape[0]
    rng = np.random.default_rng(20260401)
    picks = [rng.integers(0, n, samples // 2), [0, n - 1]]
    if mass_len:
        picks += [rng.integers(0, mass_len, samples // 2), [mass_len - 1, min(mass_len, n - 1)]]
    idx = np.unique(np.concatenate(picks)).astype(np.uint32)
    acc64, mag = ob.acc_f64_subset(before, mass_len, idx)
    got = after[idx, 4:6].astype(np.float64)
    bound = 1e-4 * np.abs(acc64) + 1e-6 * mag
    safe = np.where(bound > 0, bound, 1.0)
    ratio = np.where(bound > 0, np.abs(got - acc64) / safe, np.where(got == acc64, 0.0, np.inf))
    avx = ob.acc_avx_subset(before, mass_len, idx).astype(np.float64)
    scale = np.where(mag > 0, mag, 1.0)
    avx_ratio = np.where(bound > 0, np.abs(avx - acc64) / safe, 0.0)
    v = before[:, 2:4] + after[:, 4:6] * np.float32(dt)
    pos = before[:, 0:2] + v * np.float32(dt)
    return {"checked": int(idx.size), "worst_ratio": float(ratio.max()),
            "gpu_vs_avx_max": float((np.abs(got - avx) / scale).max()),
            "avx_vs_f64_worst_ratio": float(avx_ratio.max()),
            "integrator_bit_exact": bool(np.array_equal(after[:, 2:4], v) and np.array_equal(after[:, 0:2], pos)),
            "static_fields_equal": bool(np.array_equal(after[:, 6:8], before[:, 6:8])),
            "note": "one extra step after the timed call, outside the timed region; worst_ratio = |acc_gpu - acc_f64| / "
                    "(1e-4 |acc_f64| + 1e-6 sum|contrib|) over the sampled receivers (<= 1 = within the stated tolerance); "
                    "gpu_vs_avx_max = |acc_gpu - acc_avx| / sum|contrib|; integrator identity over all particles"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", dest="n", type=int, default=N_PARTICLES,
                    help="particles (default 2^20, the size the metric is quoted on); not --n: torchrun claims that prefix")
    ap.add_argument("--extra-particles", dest="n5", type=int, default=N_CONFIG5,
                    help="size of the second sharded workload under extra_configs (default 2^22 = BASELINE.json config 5)")
    ap.add_argument("--transport", choices=("auto", "rccl", "direct", "host"), default="auto",
                    help="N > 1: auto (default) = rccl -> direct -> host: an attempt that does not deliver a verified headline (a rank "
                         "that leaves during bring-up, a time-out, a failed self-check) is followed by a FRESH set of rank processes "
                         "over the next transport, and the line says so (transport_fallback).  rccl = in-stream ncclAllGather (the "
                         "product path); direct = no RCCL: every rank pushes its slice device-to-device into its peers' IPC-mapped "
                         "source arrays, one host barrier per step; host = slices staged through the host over the rendezvous link "
                         "(slow, but needs neither RCCL nor IPC: the transport nothing can refuse).  direct and host let several "
                         "ranks share ONE GPU, where RCCL refuses duplicate devices")
    ap.add_argument("--budget-s", type=float, default=480.0,
                    help="total wall-clock budget of the run (default 480 s, below the 600 s a driver allows the command): N > 1: "
                         "every attempt's limit, the rank link's timeout and the library's collective watchdog are carved from it and "
                         "the headline leg runs under a deadline; 1 GPU: optional legs are skipped once it is used up.  When it "
                         "expires the run still prints one line (value null if no headline was reached) and exits non-zero")
    ap.add_argument("--rendezvous", choices=("socket", "gloo"), default="socket",
                    help="N > 1: how the ranks meet on the host.  socket = a stdlib Unix-socket hub (no torch import: the run "
                         "binds /opt/rocm's HIP runtime and librccl); gloo = torch.distributed (torch's bundled runtime loads first)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-probe", action="store_true",
                    help="skip the 40 ms clock probe after the headline leg (roofline.held_clock_ghz and what follows from it)")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the parity stamps (one extra step per leg, checked against the oracle, outside the timed regions)")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline leg only: no repeats / alt_lds / extra_configs (1 GPU), no self_check / extra_configs (N GPUs)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="1 GPU: keep the repeats and the LDS-route leg but skip extra_configs (C1 / C2 / C3), so that a profiler's "
                         "per-kernel averages hold the N = 2^20 launches only (tools/profile.sh)")
    ap.add_argument("--all-massive", action="store_true",
                    help="informational N^2 run: every particle is a source (not the BASELINE.json workload)")
    ap.add_argument("--dry-run", action="store_true",
                    help="walk the multi-rank control flow (rendezvous, id broadcast, barriers, reductions, JSON keys) "
                         "without touching a GPU: no step runs and the reported value is 0")
    ap.add_argument("--leg-deadline-s", type=float, default=120.0,
                    help="N > 1: host-side deadline of every optional leg (below the library's own 180 s collective watchdog)")
    return ap.parse_args(argv)


def worker_main(args):
    """One rank: everything that touches the GPU happens in a process that runs this (and nothing else) exactly once."""

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py worker: --gpus N > 1 without WORLD_SIZE (the supervisor in main() sets it)")
        args.gpus = world
    attempt = int(os.environ.get("NB_BENCH_ATTEMPT", "0"))
    if args.transport == "auto":
        args.transport = "rccl"   # the supervisor passes an explicit transport to every attempt; a lone rank has no fallback to make
    # the run's budget as a point on CLOCK_MONOTONIC (one clock for every process of the box): handed down by the supervisor
    # for this attempt, else counted from here
    hard_deadline = float(os.environ.get("NB_BENCH_DEADLINE_MONO", 0.0)) or time.monotonic() + args.budget_s

    def time_left():
        return hard_deadline - time.monotonic()

    # stdout carries exactly one line, the JSON: RCCL prints a version banner to stdout from native code, so the
    # process' fd 1 is pointed at stderr for the duration and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # multi-process GPU work on this pool needs dmabuf IPC; RCCL's own log goes to a per-rank file so that stdout
    # keeps the one JSON line (the library's watchdog prints the file's tail if a collective never completes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sharded = world > 1 or os.environ.get("NB_HIP_FORCE_SHARDED", "0") not in ("", "0")
    if sharded:
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", f"/tmp/nbody_bench_rccl_rank{rank}_{os.getpid()}.log")

    import nbody_amd as nb  # libnbody_hip.so is loaded at the first call; aborts later if no gfx950 answers

    # ---- the line: one JSON object on rank 0, written once, whatever happens ------------------------------------
    out = {}
    out_lock = threading.Lock()
    emitted = {"done": False}

    def emit(extra_keys=None):
        with out_lock:
            if emitted["done"] or rank != 0:
                return
            emitted["done"] = True
            line = dict(out)
            if extra_keys:
                line.update(extra_keys)
            os.write(json_fd, (json.dumps(line) + "\n").encode())

    def put(key, val):
        if rank == 0:
            with out_lock:
                out[key] = val

    def emit_partial(leg):
        """A leg passed its deadline: the headline in hand goes out with the leg's name; without one, a line that says so."""
        if "value" in out:
            emit({"extras_aborted": leg})
        else:
            emit({"metric": "particle-pair interactions/sec at N=2^20", "value": None, "unit": "interactions/s", "n_gpus": world,
                  "error": f"leg '{leg}' passed its deadline before a headline was in hand (budget {args.budget_s:g} s)"})

    # every leg of a multi-rank run -- the headline included -- runs under a host-side deadline that never reaches past the
    # budget: a bring-up that blocks (an IPC open, a first barrier) ends in a line and exit 4, not in the driver's kill
    guard = LegGuard(rank, emit_partial, args.leg_deadline_s, hard_deadline - 3.0) if sharded and not args.dry_run else None
    if guard:
        guard.arm("headline (rendezvous, preflight, bring-up, timed steps)", max(5.0, time_left() - 3.0))

    # The ranks meet on the host only when launched through torch.distributed.run (also at world == 1, so that a
    # single-GPU box can rehearse the whole multi-rank flow with NB_HIP_FORCE_SHARDED=1): by default over a stdlib
    # socket hub, so that no torch -- and with it no second HIP runtime / librccl -- is in the process.
    dist = None
    torch = None
    link = None
    if "RANK" in os.environ and "MASTER_PORT" in os.environ:
        wait_s = max(5.0, time_left())
        if args.rendezvous == "gloo":
            import torch
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=wait_s))
        else:
            from nbody_amd.ranklink import RankLink

            link = RankLink(rank, world, timeout_s=wait_s, name="nbody_bench_%s_%s_a%d" % (
                os.environ["MASTER_PORT"], os.environ.get("TORCHELASTIC_RUN_ID", "none"), attempt))
        notify("rendezvous", link=link, rank=rank, world=world, transport=args.transport)
    elif world > 1:
        sys.exit("WORLD_SIZE > 1 without a torch.distributed.run rendezvous (RANK / MASTER_PORT missing)")

    def barrier():
        if dist is not None:
            dist.barrier()
        elif link is not None:
            link.barrier()

    def reduce(values, op):
        """Element-wise MIN/MAX/SUM of a list of floats over the ranks."""
        if link is not None:
            return link.reduce(values, op)
        if dist is None:
            return list(values)
        t = torch.tensor(list(values), dtype=torch.float64)
        dist.all_reduce(t, op={"max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN, "sum": dist.ReduceOp.SUM}[op])
        return [float(x) for x in t]

    def gather_bytes(mine):
        """Everybody's bytes, indexed by rank."""
        if link is not None:
            return link.allgather(bytes(mine))
        if dist is None:
            return [bytes(mine)]
        rows = [None] * world
        dist.all_gather_object(rows, bytes(mine))   # this process' own bytes, between its own ranks (gloo rendezvous only)
        return rows

    def new_unique_id():
        """rank 0 makes an RCCL unique id; the rendezvous carries its 128 bytes to the other ranks."""
        if args.dry_run:   # no GPU: /opt/rocm's ncclGetUniqueId needs one; the hand-over is what is rehearsed
            raw = bytes(range(nb.UNIQUE_ID_BYTES))
        else:
            raw = nb.comm_unique_id() if rank == 0 else bytes(nb.UNIQUE_ID_BYTES)
        raw = gather_bytes(raw)[0]
        assert len(raw) == nb.UNIQUE_ID_BYTES
        return raw

    def digests_agree(digest):
        """True when every rank's sha256 equals rank 0's."""
        return all(d == bytes(digest) for d in gather_bytes(digest))

    current_leg = {"name": "headline"}
    host_gather = None
    link_gather = None
    if sharded:
        def link_gather(rows, r, n):
            """In-place all-gather of host rows over the rendezvous (rows[r] is filled on entry)."""
            notify("gather", leg=current_leg["name"], rank=r)
            for q, row in enumerate(gather_bytes(rows[r].tobytes())):
                if q != r:
                    rows[q] = np.frombuffer(row, dtype=rows.dtype)

    if args.transport in ("host", "direct") and sharded:
        host_gather = link_gather

    def make_sim(n_, m_):
        if not sharded:
            return nb.SimPipeline(n_, m_)
        if host_gather is not None:
            return nb.SimPipeline(n_, m_, rank=rank, nranks=world, allgather=host_gather, direct=args.transport == "direct")
        return nb.SimPipeline(n_, m_, rank=rank, nranks=world, unique_id=new_unique_id())

    # ---- preflight: what every attempt writes down BEFORE the first real contact between the ranks ----------------
    flight = {"rank": rank, "transport": args.transport}

    def say_flight(**fields):
        """One `[preflight] {json}` line on stderr per stage, at once: a bring-up that dies later has left its trail with the
        supervisor (nbody_amd/launch.py keeps the lines per attempt), and rank 0 puts every rank's record on the line."""
        flight.update(fields)
        print("[preflight] " + json.dumps(dict(fields, rank=rank, transport=args.transport)), file=sys.stderr, flush=True)

    def preflight():
        if args.dry_run:
            say_flight(stage="device", dry_run=True)
            return
        visible, row = nb.preflight_peers()
        say_flight(stage="device", pci=nb.device_info().split("pci=")[-1], device=local_rank % max(visible, 1), visible_devices=visible,
                   can_access_peer=row, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
        if world > 1 and args.transport != "host":     # the host transport needs no IPC: it must not depend on the probe either
            say_flight(stage="ipc", entering="hipIpcGetMemHandle / hipIpcOpenMemHandle of the next rank's word")
            rc, handle = nb.preflight_ipc_export(0x6e620000 + rank)
            handles = gather_bytes(bytes([rc & 0xff]) + handle)
            peer = (rank + 1) % world
            if rc == 0 and handles[peer][0] == 0:
                orc, ms = nb.preflight_ipc_open(handles[peer][1:], 0x6e620000 + peer)
            else:
                orc, ms = None, None
            barrier()                                    # everybody has closed what it opened
            nb.hip_lib().nb_hip_preflight_ipc_release()
            say_flight(stage="ipc", ipc_export_rc=rc, ipc_export_error=None if rc == 0 else nb.hip_error_string(rc), ipc_open_peer=peer,
                       ipc_open_rc=orc, ipc_open_error=None if not orc else nb.hip_error_string(orc), ipc_open_ms=ms)
        if args.transport == "rccl":
            say_flight(stage="rccl", entering="ncclCommInitRank + first all-gather (pipeline creation)")

    if not args.dry_run:
        ndev = nb.device_count()
        nb.hip_lib().nb_hip_set_device(local_rank if local_rank < max(ndev, 1) else local_rank % max(ndev, 1))
    if sharded:
        preflight()
    part, mass_len = make_workload(args.n, args.all_massive)
    n = part.shape[0]

    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and not args.dry_run:
        cpu = cpu_baseline(part, mass_len)   # before the GPU legs: those then run back to back

    def timed_leg(sim, steps, warmup, dt=DT):
        """W untimed steps, then exactly K steps between barrier + device sync on both sides; max over ranks."""
        if warmup > 0:
            sim.update(warmup, dt)
        barrier()
        sim.sync()
        t0 = time.perf_counter()
        sim.update(steps, dt)   # ONE call, K steps, blocking (hipGraph chain / RCCL-stepped chain)
        sim.sync()
        barrier()
        return reduce([time.perf_counter() - t0], "max")[0]

    def sharded_detail(sim, steps):
        """Per-step kernel and all-gather device time of the last update, reduced over the ranks."""
        covered, k_ms, c_ms = sim.step_breakdown()
        per = max(covered, 1)
        k, c = k_ms / per, c_ms / per
        kmin, cmin = reduce([k, c], "min")
        kmax, cmax = reduce([k, c], "max")
        return {"steps_covered": covered, "kernel_ms_per_step": {"min": kmin, "max": kmax},
                "comm_ms_per_step": {"min": cmin, "max": cmax}}

    def comm_evidence(sim):
        info = sim.comm_info()
        up = sim.comm_bringup()
        lo = reduce([info["nranks"], info["rank"], info["device"], up["small_gather_us"], up["comm_init_ms"]], "min")
        hi = reduce([info["nranks"], info["rank"], info["device"], up["small_gather_us"], up["comm_init_ms"]], "max")
        sm = reduce([info["rank"], 1.0 if info["owns_comm"] else 0.0], "sum")
        return {
            "nranks_reported": {"min": int(lo[0]), "max": int(hi[0])},       # ncclCommCount on every rank
            "user_ranks": {"min": int(lo[1]), "max": int(hi[1]), "sum": int(sm[0])},  # ncclCommUserRank: 0..P-1, sum P(P-1)/2
            "devices": {"min": int(lo[2]), "max": int(hi[2])},              # ncclCommCuDevice
            "ranks_with_communicator": int(sm[1]),
            "version": info["rccl_version"],
            "lib": info["rccl_lib"],
            "first_gather_ms_rank0": info["first_gather_ms"],
            # bring-up, measured at creation (nb_hip_comm_bringup): the warm 8-byte all-gather is the fixed cost of the per-step
            # gather -- the number that replaces the 50 us ASSUMED in every S2 / S4 / S8 prediction
            "comm_init_ms": {"min": lo[4], "max": hi[4]},
            "small_gather_us": {"min": lo[3], "max": hi[3]},
        }

    # ---- the headline leg ---------------------------------------------------------------------------------------
    extras = {}
    steps_done = 0
    sim = None
    if args.dry_run:
        uid = new_unique_id() if sharded else None
        assert uid is None or uid == bytes(range(nb.UNIQUE_ID_BYTES))
        plan = nb.shard_plan(n, mass_len, rank, world)
        assert plan["mass_count"] + plan["zero_count"] > 0 or n < world
        barrier()
        t0 = time.perf_counter()
        barrier()
        elapsed = reduce([max(time.perf_counter() - t0, 1e-9)], "max")[0]
        kernel_ms, launches, finish_launches, clock = 0.0, 0, 0, None
        shape, info = nb.plan_launch(plan["mass_count"] + plan["zero_count"], plan["src_padded"]), "dry-run"
        if sharded:
            extras["rccl"] = {"nranks_reported": {"min": None, "max": None}, "user_ranks": None, "devices": None,
                              "ranks_with_communicator": 0, "version": None, "lib": None, "first_gather_ms_rank0": None,
                              "comm_init_ms": None, "small_gather_us": None}
            zero = {"min": 0.0, "max": 0.0}
            extras["comm_ms_per_step"], extras["kernel_ms_per_step"] = dict(zero), dict(zero)
        runtime = None
    else:
        sim = make_sim(n, mass_len)
        if sharded and args.transport == "rccl":
            say_flight(stage="rccl", **{k: v for k, v in sim.comm_bringup().items() if k != "owns_comm"})
        if not sharded:
            sim.configure(graph=1)   # the K-step chain runs as a hipGraph on its first use, built inside the timed call
        sim.set_data(part)           # H2D + SoA split: outside the timed region
        elapsed = timed_leg(sim, args.steps, args.warmup)
        steps_done = args.warmup + args.steps
        kernel_ms, launches = sim.last_step_ms()
        clock = None   # single GPU: the clock probe runs once the line below is in hand (so does the parity stamp)
        finish_launches = sim.finish_launches()
        shape = sim.launch_shape()
        info = nb.device_info()
        runtime = {"hip_runtime_version": int(nb.hip_lib().nb_hip_runtime_version()),
                   "torch_imported_first": torch is not None,
                   "rendezvous": "gloo" if dist is not None else ("socket hub" if link is not None else None)}
        if sharded:
            extras["rccl"] = comm_evidence(sim)
            d = sharded_detail(sim, args.steps)
            extras["comm_ms_per_step"], extras["kernel_ms_per_step"] = d["comm_ms_per_step"], d["kernel_ms_per_step"]
    if sharded:
        # every rank's bring-up record, on the line (the supervisor also holds the per-stage trail of attempts that died)
        extras["preflight"] = [json.loads(b.decode()) for b in gather_bytes(json.dumps(flight).encode())]

    # ---- the JSON dict: complete from here on; later legs only add keys -----------------------------------------
    if rank == 0:
        interactions = float(n) * float(mass_len) * args.steps
        value = 0.0 if args.dry_run else interactions / elapsed
        # dominant kernel: the step kernel; algorithmic flops per launch = interactions per launch * 14
        per_launch_s = (kernel_ms * 1e-3) / max(launches, 1)
        launch_interactions = float(n) * float(mass_len) / world * (args.steps / max(launches, 1))
        achieved_tflops = launch_interactions * FLOP_PER_INTERACTION / per_launch_s / 1e12 if per_launch_s > 0 else 0.0
        passes = max(launches // max(args.steps, 1), 1)
        unmeasured = args.all_massive or world > 1 or args.dry_run
        traffic, traffic_note = (None, "not measured for this workload") if unmeasured else pmc_traffic(n, shape, passes)
        roof = {
            "bound": "valu",  # fp32 vector ALU (rsq + fma); neither HBM nor MFMA bounds this path
            "achieved": achieved_tflops,
            "peak": PEAK_FP32_VECTOR_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved_tflops / PEAK_FP32_VECTOR_TFLOPS,
            # the same fraction from the wall clock of the K timed steps instead of the kernel's HIP events (= value x 14 /
            # peak per GPU): the two differ by whatever a step spends outside its step kernels
            "roofline_frac_from_wall": value * FLOP_PER_INTERACTION / (PEAK_FP32_VECTOR_TFLOPS * 1e12 * world),
            "traffic": traffic,
            "traffic_measured_in_this_run": False,   # PMC passes cannot run inside the timed process: looked up, see traffic_note
            "traffic_note": traffic_note,
            "traffic_parts": None if unmeasured else pmc_traffic_parts(n, shape, passes),
            "traffic_algorithmic": algorithmic_bytes_per_launch(n, mass_len, passes) if world == 1 else None,
            "flop_per_interaction": FLOP_PER_INTERACTION,
            "kernel_ms_per_launch": per_launch_s * 1e3,
            "launches": launches,
            "finish_launches": finish_launches,
            **held_clock_fields(clock, None, per_launch_s, launch_interactions, info, achieved_tflops),
            "kernel_ms_note": ("HIP events on the launch stream around the whole chain / step-kernel launches"
                               + ("; each interval also holds one O(N) finish kernel (~9 us at 2^20) per step launch"
                                  if finish_launches else "")
                               + ("; sharded: the interval includes the all-gathers, see kernel_ms_per_step" if sharded else "")),
        }
        with out_lock:
            out.update({
                "metric": "particle-pair interactions/sec at N=2^20",
                "value": value,
                "unit": "interactions/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3,
                "steps_per_sec": args.steps / elapsed,
                "higher_is_better": True,
                "scaling": "strong",
                "vs_baseline": None,
                "dtype": "f32",
                "data": "synthetic",
                "config": {
                    "workload": f"srand(11037) MakeGalaxies({n}, 2) (galaxy.h ICs)"
                                f"{', massless half given NP_R_TO_M(radius) mass (all-massive N^2 run)' if args.all_massive else ''}"
                                f", partitioned; N={n}, mass_len={mass_len}, "
                                f"dt={DT}; {n * mass_len:.4g} interactions/step; one PerformSimUpdate({args.steps}) call",
                    "parallelism": (f"receivers sharded N/{world} per GPU, all-gather of source positions per step"
                                    + (" by direct device-to-device pushes (no RCCL)" if args.transport == "direct" else
                                       " staged through the host over the rendezvous link (no RCCL, no IPC)" if host_gather else ""))
                                   if world > 1 else "single GPU",
                    "kernel": shape,
                    "device": info,
                },
                "roofline": roof,
                "runtime": runtime,
            })
            if sharded:
                # ncclCommCount as seen by every rank's communicator -- null when a rank holds none (host transport, dry run)
                out["rccl_nranks"] = (extras["rccl"]["nranks_reported"]["min"]
                                      if extras["rccl"]["ranks_with_communicator"] == world else None)
                out["transport"] = ("direct (device-to-device pushes into IPC-mapped peers, one barrier per step over the rendezvous link)"
                                    if args.transport == "direct" else
                                    "host (all-gather over the rendezvous link through page-locked staging)" if host_gather
                                    else "rccl (in-stream ncclAllGather)")
            out.update(extras)
            if cpu is not None:
                out["cpu_baseline"] = cpu

    # Single GPU: from here on every further leg -- clock probe, parity stamp, repeats, the clock-sampler leg, the LDS
    # route, extra_configs -- runs with the line in hand: the library's error convention is abort(), and a fatal signal
    # inside any of them still writes the headline (plus "extras_aborted": which leg) through the C-level handler.
    solo_gasp = LastGasp(json_fd) if (rank == 0 and not sharded and not args.dry_run) else None
    skipped_legs = []

    def solo_leg(name, need_s=0.0):
        """Arms the last-gasp line for the leg and says whether the budget still holds it (need_s: what the leg takes)."""
        if time_left() < need_s:
            skipped_legs.append(name)
            put("legs_skipped_for_budget", list(skipped_legs))
            return False
        if solo_gasp:
            with out_lock:
                solo_gasp.arm((json.dumps(dict(out, extras_aborted=f"{name} (fatal signal)")) + "\n").encode())
        notify("leg", leg=name, rank=rank, solo=True)
        return True

    if not sharded and not args.dry_run and rank == 0:
        if not args.no_clock_probe and solo_leg("clock probe", 2.0):
            # the clock the chip holds for the interaction statement alone, asked right after the timed steps and outside
            # them: a separate probe kernel (include/nbody_hip.h nb_hip_probe_clock); the product kernels carry no stamps
            try:
                clock = nb.probe_clock(40.0)
            except Exception as e:  # pragma: no cover - diagnostic only
                clock = {"error": str(e)}
            with out_lock:
                out["roofline"].update(held_clock_fields(clock, None, per_launch_s, launch_interactions, info, achieved_tflops))
        if not args.no_parity and solo_leg("parity stamp", 10.0):
            put("parity", parity_stamp(sim, mass_len))   # after the timed call, outside it

    # ---- optional legs ------------------------------------------------------------------------------------------
    if sharded:
        gasp = LastGasp(json_fd) if (rank == 0 and not args.dry_run) else None

        def leg(name, need_s=0.0):
            """Arms the deadline and the last-gasp line for the leg; False (on every rank alike) when the budget no longer
            holds what the leg needs -- it is then listed under legs_skipped_for_budget instead of being started."""
            if need_s and reduce([time_left()], "min")[0] < need_s + 5.0:
                skipped_legs.append(name)
                put("legs_skipped_for_budget", list(skipped_legs))
                return False
            current_leg["name"] = name
            if guard:
                guard.arm(name)
            if gasp:
                with out_lock:
                    gasp.arm((json.dumps(dict(out, extras_aborted=f"{name} (fatal signal)")) + "\n").encode())
            return True

        def done_with_legs():
            if guard:
                guard.disarm()
            if gasp:
                gasp.disarm()

        single_gpu_state = {}

        def self_check(pipeline, steps_run):
            """NOT optional for any N > 1 headline (--no-extras keeps it): every rank must hold the same full state, and it
            must be the single-GPU state of the same steps -- the check that would catch a stale or torn exchange."""
            if args.dry_run:
                check = {"ranks_agree": digests_agree(hashlib.sha256(part.tobytes()).digest()), "vs_single_gpu_rel_l2_pos": None}
                notify("self_check", check=check, transport=args.transport)
                return check
            got = pipeline.get_data()  # collective
            check = {"ranks_agree": digests_agree(hashlib.sha256(got.tobytes()).digest()), "steps": steps_run}
            notify("self_check", check=check, transport=args.transport)
            if rank == 0:
                if steps_run not in single_gpu_state:
                    one = nb.SimPipeline(n, mass_len)
                    one.set_data(part)
                    one.update(steps_run, DT)
                    single_gpu_state[steps_run] = one.get_data()
                    one.close()
                want = single_gpu_state[steps_run]
                dp = (got[:, 0:2].astype(np.float64) - want[:, 0:2]).ravel()
                check["vs_single_gpu_rel_l2_pos"] = float(np.sqrt(dp @ dp) / np.linalg.norm(want[:, 0:2].astype(np.float64)))
                check["vs_single_gpu_max_abs_pos"] = float(np.abs(dp).max())
                check["static_fields_equal"] = bool(np.array_equal(got[:, 6:8], want[:, 6:8]))
                # the sharded sum differs from the single-GPU one only in the order M terms are added (1e-4: DESIGN.md section 5)
                check["ok"] = bool(check["ranks_agree"] and check["static_fields_equal"] and check["vs_single_gpu_rel_l2_pos"] <= 1e-4)
            # which physical devices took part: PCI addresses (ordinals can all read 0 when every rank sees one GPU)
            seen = sorted(set(gather_bytes(nb.device_info().split("pci=")[-1].encode())))
            check["devices"] = [d.decode() for d in seen]
            check["crossed_devices"] = len(seen) > 1
            barrier()
            return check

        leg("self_check")
        put("self_check", self_check(sim, steps_done))
        if args.dry_run:
            part5, m5 = make_workload(args.n5)
            _ = new_unique_id()
            p5 = nb.shard_plan(part5.shape[0], m5, rank, world)
            assert p5["src_padded"] >= m5
            put("extra_configs", [] if args.no_extras else [
                _extra_entry(n, mass_len, 1, 0, args.steps, 0.0, None, world),
                _extra_entry(part5.shape[0], m5, 0, 0, 3, 0.0, None, world),
                _extra_entry(part5.shape[0], m5, 1, 0, 3, 0.0, None, world),
            ] + ([dict(_extra_entry(n, mass_len, 0, 0, args.steps, 0.0, None, world), transport="direct (dry run)")]
                 if host_gather is None else [])
              + [_extra_entry(n, mass_len, 0, 1, args.steps, 0.0, None, world)])
        elif args.no_extras:
            done_with_legs()
            sim.close()
            sim = None
        else:
            extra = []
            put("extra_configs", extra)   # the list grows in place: a deadline line carries the legs that finished
            step_s = elapsed / args.steps
            # the overlapped step on the same pipeline
            if leg("overlap", 2 * (args.steps + 1) * step_s):
                sim.configure(overlap=1)
                e1 = timed_leg(sim, args.steps, 1)
                extra.append(_extra_entry(n, mass_len, 1, 0, args.steps, e1, sharded_detail(sim, args.steps), world))
                sim.configure(overlap=0)
            # BASELINE.json config 5: N = 2^22, plain and overlapped (own communicator: a second ncclCommInitRank)
            scale5 = (args.n5 / float(n)) ** 2
            if leg("config5", 20.0 + 2 * 8 * step_s * scale5):
                part5, m5 = make_workload(args.n5)
                sim5 = make_sim(part5.shape[0], m5)
                sim5.set_data(part5)
                for ov in (0, 1):
                    leg("config5")
                    sim5.configure(overlap=ov)
                    e5 = timed_leg(sim5, 3, 1)
                    extra.append(_extra_entry(part5.shape[0], m5, ov, 0, 3, e5, sharded_detail(sim5, 3), world))
                sim5.close()
            # the direct exchange (no RCCL; slices pushed device-to-device into IPC-mapped peers,
            # one barrier per step over the rendezvous link) on the headline workload, for an RCCL-vs-direct comparison
            # from the same command -- only when the run's own transport is RCCL (otherwise the legs above were it)
            if host_gather is None and link_gather is not None and leg("direct", 10.0 + 3 * (args.steps + 1) * step_s * world):
                simd = nb.SimPipeline(n, mass_len, rank=rank, nranks=world, allgather=link_gather, direct=True)
                simd.set_data(part)
                ed = timed_leg(simd, args.steps, 1)
                entry = _extra_entry(n, mass_len, 0, 0, args.steps, ed, sharded_detail(simd, args.steps), world)
                entry["transport"] = "direct (device-to-device pushes into IPC-mapped peers, one host barrier per step)"
                # checked like the headline: same bytes on every rank, and the single-GPU state of the same steps
                entry["self_check"] = self_check(simd, 1 + args.steps)
                entry["cross_device_parity"] = ("pinned by this run: the ranks sat on different devices and the self-check passed"
                                                if entry["self_check"]["crossed_devices"] and entry["self_check"].get("ok", True) else
                                                "unpinned across devices: every rank of this run sat on the same device"
                                                if not entry["self_check"]["crossed_devices"] else "FAILED across devices")
                extra.append(entry)
                simd.close()
            # LAST, because it is the least-travelled path of the stack and a stall here must not cost the legs above:
            # north star: "multi-step chains are captured as hipGraph" -- the {kernel, all-gather} x K chain captured
            # from the stream and replayed (RCCL inside stream capture; a host callback cannot be captured)
            if host_gather is not None:
                extra.append(dict(_extra_entry(n, mass_len, 0, 1, args.steps, 0.0, None, world),
                                  skipped="host / direct transport: a host callback cannot run inside a captured graph"))
            elif leg("sharded_graph", 10.0 + 3 * args.steps * step_s):
                sim.configure(sharded_graph=1)
                eg = timed_leg(sim, args.steps, args.steps)   # the warm-up call captures and instantiates the chain
                entry = _extra_entry(n, mass_len, 0, 1, args.steps, eg, None, world)
                entry["graph_stats"] = sim.graph_stats()
                extra.append(entry)
                sim.configure(sharded_graph=0)
            sim.close()
            sim = None
            done_with_legs()
    elif not sharded and not args.no_extras and not args.dry_run:
        # same K steps twice more (run-to-run spread), then the LDS-tile route of the north star on the same chain
        if solo_leg("repeats", 3 * elapsed):
            put("repeat_ms_per_step", [timed_leg(sim, args.steps, 0) / args.steps * 1e3 for _ in range(2)])
        if not args.no_clock_probe and rank == 0 and solo_leg("clock sampler leg", 2 * elapsed + 2.0):
            # the clock the chip holds UNDER THE STEP KERNEL: the same K steps once more with the sampler running beside them
            try:
                # the sampler's own bound (the library clamps to it too): a leg longer than that is sampled over its first 20 s
                nb.clock_sampler_begin(0.5, min(nb.CLOCK_SAMPLER_MAX_MS, 3.0 * elapsed * 1e3 + 500.0))
                e_clk = timed_leg(sim, args.steps, 0)
                clk_ms, clk_launches = sim.last_step_ms()
                sampled = nb.clock_sampler_end()
                clk_s = clk_ms * 1e-3 / max(clk_launches, 1)
                sampled["leg"] = {"ms_per_step": e_clk / args.steps * 1e3, "kernel_ms_per_launch": clk_s * 1e3,
                                  "cycles_per_wave_interaction": clk_s * sampled["clock_ghz"] * 1e9 * 4 * device_cus(info)
                                  / (launch_interactions / 64.0) if sampled.get("clock_ghz") else None}
                if e_clk * 1e3 > nb.CLOCK_SAMPLER_MAX_MS:
                    sampled["note"] = f"the leg took {e_clk:.1f} s: sampled over its first {nb.CLOCK_SAMPLER_MAX_MS / 1e3:g} s only"
                with out_lock:
                    out["roofline"].update(held_clock_fields(clock, sampled, per_launch_s, launch_interactions, info, achieved_tflops))
            except Exception as e:  # pragma: no cover - diagnostic only
                put("clock_sampler_error", str(e))
        if solo_leg("alt_lds", 2 * elapsed + 2.0):
            sim.configure(variant=0)
            e_lds = timed_leg(sim, args.steps, 2)
            lds_ms, lds_launches = sim.last_step_ms()
            lds_s = lds_ms * 1e-3 / max(lds_launches, 1)
            lds_tf = float(n) * float(mass_len) * (args.steps / max(lds_launches, 1)) * FLOP_PER_INTERACTION / lds_s / 1e12
            with out_lock:
                out["roofline"]["alt_lds"] = {
                    "note": "same K steps through the LDS-tile source route (north star's design; variant=0), bit-identical "
                            "results, timed after the headline leg; why it trails: profiles/r03_routes_pmc.txt",
                    "ms_per_step": e_lds / args.steps * 1e3, "value": float(n) * float(mass_len) * args.steps / e_lds,
                    "kernel_ms_per_launch": lds_s * 1e3, "achieved": lds_tf,
                    "frac": lds_tf / PEAK_FP32_VECTOR_TFLOPS, "kernel": sim.launch_shape()}
        sim.close()
        sim = None
        if not args.all_massive and args.n == N_PARTICLES and not args.no_extra_configs:
            configs = []
            put("extra_configs", configs)
            if solo_leg("extra_configs C2/C3/N2/C1", 30.0):
                configs.extend(single_gpu_configs(nb, stamp=not args.no_parity))
            if solo_leg("extra_configs S2/S4/S8/C5S8", 40.0):
                try:
                    configs.extend(shard_scaling_configs(nb, elapsed / args.steps * 1e3, stamp=not args.no_parity, n5=args.n5))
                except Exception as e:  # pragma: no cover - diagnostic only
                    configs.append({"config": "S2/S4/S8/C5S8", "error": str(e)})
    if sim is not None:
        sim.close()
    if solo_gasp:
        solo_gasp.disarm()
    if guard:
        guard.disarm()

    emit()

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if link is not None:
        link.barrier()
        link.close()


def single_gpu_configs(nb, stamp=True):
    """BASELINE.json's other single-GPU configurations, in the shape of the reference harness (src/bench.c:21-35), and
    the all-massive (N^2) run SURVEY.md 8d asks to report beside the headline."""
    out = []

    def rate(n, m, sec_per_step):
        v = float(n) * float(m) / sec_per_step
        return {"ms_per_step": sec_per_step * 1e3, "value": v, "unit": "interactions/s",
                "roofline_frac": v * FLOP_PER_INTERACTION / (PEAK_FP32_VECTOR_TFLOPS * 1e12)}

    # C2: N = 65 536, one kernel per step; 10 warm-up steps, then ONE 100-step call, library defaults (what
    # nbody-bench's single timed call gets: plain launches the first time a chain length is seen, bench.c:30-33)
    part, m = make_workload(N_CONFIG2)
    n = part.shape[0]
    sim = nb.SimPipeline(n, m)
    sim.set_data(part)
    sim.update(10, DT)
    t0 = time.perf_counter()
    sim.update(100, DT)
    first = (time.perf_counter() - t0) / 100
    k_ms, launches = sim.last_step_ms()
    later = []
    for _ in range(3):   # from its second use the same chain length replays as a cached hipGraph
        t0 = time.perf_counter()
        sim.update(100, DT)
        later.append((time.perf_counter() - t0) / 100)
    entry = {"config": "C2", "workload": f"srand(11037) MakeGalaxies({n}, 2), N={n}, mass_len={m}, dt={DT}; 10 warm-up steps, "
                                         "one PerformSimUpdate(100) call, default knobs",
             "steps": 100, "kernel": sim.launch_shape(), "kernel_ms_per_launch": k_ms / max(launches, 1),
             "launches": launches, "graph_stats": sim.graph_stats()}
    entry.update(rate(n, m, first))
    entry["later_calls"] = dict(rate(n, m, min(later)), note="fastest of 3 further 100-step calls (cached hipGraph replays)")
    if stamp:
        entry["parity"] = parity_stamp(sim, m)
    # BASELINE.json words config 2 as "single LDS-tiled force+integrate kernel": the same calls through the LDS-tile source route
    # (variant = 0; bit-identical results, the route the north star names) beside the default scalar-cache route above
    sim.configure(variant=0)
    sim.update(100, DT)
    lds = []
    for _ in range(3):
        t0 = time.perf_counter()
        sim.update(100, DT)
        lds.append((time.perf_counter() - t0) / 100)
    entry["lds_route"] = dict(rate(n, m, min(lds)), kernel=sim.launch_shape(), note="fastest of 3 100-step calls through the LDS-tile route")
    if stamp:
        entry["lds_route"]["parity"] = parity_stamp(sim, m)
    sim.close()
    out.append(entry)

    # C3: N = 262 144, the K-step chain as a hipGraph: built by a warm-up call, replayed at dt, then the SAME cached
    # chain at dt/2 (the step size lives in device memory: one 4-byte upload, no rebuild, no node patched)
    part, m = make_workload(N_CONFIG3)
    n = part.shape[0]
    sim = nb.SimPipeline(n, m)
    sim.configure(graph=1)
    sim.set_data(part)
    k = 20
    sim.update(k, DT)          # builds + instantiates the 20-step chain
    g0 = sim.graph_stats()
    t0 = time.perf_counter()
    sim.update(k, DT)
    at_dt = (time.perf_counter() - t0) / k
    k_ms, launches = sim.last_step_ms()
    t0 = time.perf_counter()
    sim.update(k, DT / 2)
    at_half = (time.perf_counter() - t0) / k
    g1 = sim.graph_stats()
    entry = {"config": "C3", "workload": f"srand(11037) MakeGalaxies({n}, 2), N={n}, mass_len={m}; one cached {k}-step hipGraph "
                                         f"chain replayed at dt={DT}, then at dt={DT / 2}",
             "steps": k, "kernel": sim.launch_shape(), "kernel_ms_per_launch": k_ms / max(launches, 1), "launches": launches,
             "graph_stats_before": g0, "graph_stats_after": g1,
             "chain_rebuilt_for_new_dt": g1["cached"] != g0["cached"], "dt_uploads_for_new_dt": g1["dt_uploads"] - g0["dt_uploads"]}
    entry.update(rate(n, m, at_dt))
    entry["dt_halved"] = rate(n, m, at_half)
    if stamp:
        entry["parity"] = parity_stamp(sim, m, dt=DT / 2)
    sim.close()
    out.append(entry)

    # N2: the all-massive run (SURVEY.md 8d "also report the all-massive equivalent"): the same universe with the massless
    # half given the mass galaxy.h would give a body of its radius, so every particle is a source: N x N interactions
    part, m = make_workload(N_PARTICLES, all_massive=True)
    n = part.shape[0]
    sim = nb.SimPipeline(n, m)
    sim.configure(graph=1)
    sim.set_data(part)
    k = 4
    sim.update(1, DT)
    t0 = time.perf_counter()
    sim.update(k, DT)
    sec = (time.perf_counter() - t0) / k
    k_ms, launches = sim.last_step_ms()
    entry = {"config": "N2", "workload": f"srand(11037) MakeGalaxies({n}, 2), massless half given NP_R_TO_M(radius) mass: N={n}, "
                                         f"mass_len={m}, dt={DT}; {float(n) * m:.4g} interactions/step; one PerformSimUpdate({k}) call",
             "steps": k, "kernel": sim.launch_shape(), "kernel_ms_per_launch": k_ms / max(launches, 1), "launches": launches}
    entry.update(rate(n, m, sec))
    if stamp:
        entry["parity"] = parity_stamp(sim, m)
    sim.close()
    out.append(entry)

    # C1: N = 4 096 on the CPU path through the product's nbody-bench (plumbing; no GPU involved)
    exe = os.path.join(ROOT, "nbody_amd", "lib", "nbody-bench")
    try:
        env = dict(os.environ, OMP_NUM_THREADS=str(host_cores()))
        r = subprocess.run([exe, "--cpu", "--n", "4096", "--steps", "100"], env=env, capture_output=True, text=True, timeout=120)
        row = [line.split() for line in r.stdout.splitlines() if line.split() and line.split()[0] == "4096"][0]
        out.append({"config": "C1", "workload": "nbody-bench --cpu --n 4096 --steps 100 (srand(11037) MakeGalaxies(4096, 2), dt=1, "
                                                "10 warm-up + 100 timed steps, UpdateWorld_CPU = AVX + OpenMP)",
                    "steps": 100, "us_per_step": float(row[1]), "value": float(row[2]), "unit": "interactions/s",
                    "cores": host_cores()})
    except Exception as e:  # pragma: no cover - diagnostic only
        out.append({"config": "C1", "error": str(e)})
    return out


def shard_leg(nb, name, part, m, ranks, steps, t1_ms=None, stamp=True):
    """One rank's share of a sharded step, timed on ONE GPU: all `ranks` shards of the world live in this process
    (nb_hip_local_group_*: the same shard plan, kernels, launch shape and mirror / gather layout as the RCCL path, the
    exchange replaced by device-to-device copies on the same device) and advance together; every member's kernels carry
    their own HIP event pairs, so the entry reports what ONE rank's N/P receivers x all M sources cost -- the compute half
    of the 1/2/4/8 curve -- beside the estimated gather."""
    n = part.shape[0]
    grp = nb.LocalShardGroup(n, m, ranks)
    grp.set_data(part)
    grp.step(1, DT)   # warm-up (first launch of this shape, parts buffer)
    t0 = time.perf_counter()
    grp.step(steps, DT)
    wall = (time.perf_counter() - t0) / steps
    kernel, push = [], []
    for mem in grp.members:
        covered, k_ms, c_ms = mem.step_breakdown()
        kernel.append(k_ms / max(covered, 1))
        push.append(c_ms / max(covered, 1))
    plan = nb.shard_plan(n, m, 0, ranks)
    shard_ms = max(kernel)
    gather_ms = gather_estimate_ms(plan["mass_chunk"], ranks)
    entry = {
        "config": name,
        "workload": f"srand(11037) MakeGalaxies({n}, 2), N={n}, mass_len={m}, dt={DT}: ONE rank's step of a {ranks}-way sharded run "
                    f"({plan['mass_count'] + plan['zero_count']} receivers x all {plan['src_padded']} gathered sources), plain launches, "
                    f"timed for every one of the {ranks} shards on one GPU (local group: the exchange is a device copy, not RCCL)",
        "ranks": ranks, "steps": steps,
        "kernel": grp.members[0].launch_shape(),
        "shard_kernel_ms_per_step": {"min": min(kernel), "max": max(kernel), "mean": sum(kernel) / len(kernel)},
        "local_push_ms_per_step": {"min": min(push), "max": max(push)},
        "all_shards_wall_ms_per_step": wall * 1e3,
        "interactions_per_rank_step": float(plan["mass_count"] + plan["zero_count"]) * float(m),
        "roofline_frac_per_rank": (float(plan["mass_count"] + plan["zero_count"]) * float(m) * FLOP_PER_INTERACTION
                                   / (shard_ms * 1e-3) / (PEAK_FP32_VECTOR_TFLOPS * 1e12)),
        "gather_estimate_ms": gather_ms,
        "gather_estimate_source": f"{plan['mass_chunk']} float2 per rank over its own xGMI link at {XGMI_LINK_GBS:g} GB/s (SURVEY.md 8e: direct "
                                  f"all-gather, one slice per link) + {GATHER_LATENCY_ASSUMED_MS:g} ms assumed fixed latency; NOT measured: "
                                  "no run of this repo has crossed xGMI",
        "predicted_steps_per_sec": 1.0 / ((shard_ms + gather_ms) * 1e-3),
        "predicted_interactions_per_sec": float(n) * float(m) / ((shard_ms + gather_ms) * 1e-3),
    }
    if t1_ms is not None:
        entry["single_gpu_ms_per_step"] = t1_ms
        entry["compute_scaling_efficiency"] = t1_ms / (ranks * shard_ms)
        entry["predicted_scaling_efficiency"] = t1_ms / (ranks * (shard_ms + gather_ms))
    if stamp:
        entry["parity"] = parity_stamp(grp, m)
    if ranks > 1:
        # What would the OVERLAPPED step cost this rank?  Own-shard kernel + remote-shards kernel instead of one: the sum of the
        # two against the one above is what overlap must win back by hiding the gather.  So the first measured plain
        # comm_ms_per_step of a real multi-GPU run decides the sharded default by comparison with ONE number (DESIGN.md
        # section 4): overlap pays when the gather it hides is longer than overlap_break_even_gather_ms.
        for mem in grp.members:
            mem.configure(overlap=1)
        grp.step(1, DT)
        grp.step(steps, DT)
        both = []
        for mem in grp.members:
            covered, k_ms, _ = mem.step_breakdown()
            both.append(k_ms / max(covered, 1))
        entry["overlap_kernels_ms_per_step"] = {"min": min(both), "max": max(both), "mean": sum(both) / len(both)}
        entry["overlap_break_even_gather_ms"] = max(both) - shard_ms
        entry["overlap_note"] = ("own-shard + remote-shards kernels of the overlapped step, summed, minus the one kernel of the plain step: "
                                 "what hiding the gather must win back; compare with a real run's plain comm_ms_per_step")
    grp.close()
    return entry


def shard_scaling_configs(nb, t1_ms, stamp=True, n5=N_CONFIG5):
    """extra_configs S2 / S4 / S8 (the headline workload cut 2 / 4 / 8 ways) and C5S8 (BASELINE.json config 5: N = 2^22, 8 ways):
    the per-rank shard step measured on this one GPU, each with a parity stamp against float64."""
    out = []
    part, m = make_workload(N_PARTICLES)
    for ranks in (2, 4, 8):
        out.append(shard_leg(nb, f"S{ranks}", part, m, ranks, 3, t1_ms=t1_ms, stamp=stamp))
    part5, m5 = make_workload(n5)
    out.append(shard_leg(nb, "C5S8", part5, m5, 8, 2, stamp=stamp))
    return out


def _extra_entry(n, m, overlap, sharded_graph, steps, elapsed, detail, world):
    e = {
        "workload": f"srand(11037) MakeGalaxies({n}, 2), N={n}, mass_len={m}, dt={DT}, N/{world} receivers per GPU",
        "overlap": overlap,   # 1 = own-shard kernel runs while the other shards' positions are still being gathered
        "sharded_graph": sharded_graph,   # 1 = the {kernel, all-gather} x K chain captured as a hipGraph and replayed
        "steps": steps,
        "ms_per_step": elapsed / steps * 1e3,
        "steps_per_sec": steps / elapsed if elapsed > 0 else 0.0,
        "value": float(n) * float(m) * steps / elapsed if elapsed > 0 else 0.0,
        "unit": "interactions/s",
    }
    if detail is not None:
        e["kernel_ms_per_step"] = detail["kernel_ms_per_step"]
        e["comm_ms_per_step"] = detail["comm_ms_per_step"]
    else:
        e["kernel_ms_per_step"] = e["comm_ms_per_step"] = {"min": 0.0, "max": 0.0}
    return e


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if os.environ.get("NB_BENCH_WORKER") == "1" or (args.gpus <= 1 and not launched):
        if os.environ.get("NB_BENCH_REHEARSE"):     # failure rehearsals live with the tests, not here (tests/bench_rehearsal.py)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import bench_rehearsal
            bench_rehearsal.install(sys.modules[__name__])
        worker_main(args)     # one rank; the only place a GPU is touched
        return 0
    return supervise(args, argv, os.path.abspath(__file__))   # GPU-free: nbody_amd/launch.py


if __name__ == "__main__":
    sys.exit(main())
