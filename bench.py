#!/usr/bin/env python3
"""bench.py -- the hot path's headline number: particle-pair interactions/s at N = 2^20 on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one
process per GPU through torch.distributed.run.  One JSON line on rank 0.

Workload (BASELINE.json metric): srand(11037); MakeGalaxies(2^20, 2) -- the reference
bench's universe (src/bench.c:42,53) at the size the metric is quoted on -- partitioned by CreateWorld,
dt = 0.01.  A "step" is one force + integrate pass over all N receivers against all mass_len sources.
Interactions per step = N * mass_len (what the reference kernels evaluate, particle_cs.glsl:30,35).
K steps run as ONE PerformSimUpdate(K) call, like the reference harness' update(w, dt, 100) (bench.c:30-33);
particles are resident in HBM before the timed region (SetSimulationData is outside it).

N > 1: strong scaling -- the same 2^20 particles, N/P receivers per GPU, all-gather of source positions
per step over RCCL inside the library; torch.distributed (gloo) only carries the rendezvous, the barrier
and the max-over-ranks of the time.

The oracle (oracle/) is used here ONLY for the cpu_baseline leg.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_INTERACTION = 14        # reference op count, sim_cpu.c:169-188
PEAK_FP32_VECTOR_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (vector)"
N_PARTICLES = 1 << 20
DT = 0.01


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


def cpu_baseline(part, mass_len, budget_s=12.0):
    """Reference AVX path timed on this box's host cores over a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))  # the GPU box gives one GPU a 16-CPU share
    n = part.shape[0]
    # ~2e9 interactions/s/core: size the receiver sample for about budget_s seconds
    recv = int(budget_s * 2.0e9 * cores / max(mass_len, 1))
    recv = max(64 * cores, min(n, recv // (20 * cores) * (20 * cores)))
    kind, threads = "port", cores
    sec = None
    if os.path.exists(ob.REF_CPU_SO):
        try:
            sec = _time_reference_packedupdate(ob.REF_CPU_SO, part, mass_len, recv, cores)
            kind = "reference"
        except Exception as e:  # pragma: no cover - diagnostic only
            print(f"[bench] reference CPU leg failed ({e}); using the port", file=sys.stderr)
    one = None
    if sec is None:
        sec, threads, _ = ob.time_avx_sample(part, mass_len, 0, recv, dt=DT, threads=cores)
    else:
        # SURVEY.md 8d also asks for the 1-thread figure: ~2 s of the same loop on one core
        recv1 = max(64, min(n, int(2.0 * 2.0e9 / max(mass_len, 1))))
        one = recv1 * mass_len / _time_reference_packedupdate(ob.REF_CPU_SO, part, mass_len, recv1, 1)
    return {
        "value": recv * mass_len / sec,
        "unit": "interactions/s",
        "cores": threads,
        "kind": kind,
        "cpu_model": _cpu_model(),
        "value_1_thread": one,
        "sample": f"{recv} of {n} receivers x all {mass_len} sources, one step, AVX (-mavx, no FMA) + {threads} threads"
                  f" ({sec:.2f} s); whole step would take ~{sec * n / recv:.0f} s",
    }


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def _time_reference_packedupdate(so, part, mass_len, recv, cores):
    """oracle/_ref = the reference's own sim_cpu.c, driven like world.c:101-107 from `cores` threads."""
    from concurrent.futures import ThreadPoolExecutor

    ref = C.CDLL(so)
    ref.AllocPackArray.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32]
    ref.PackParticles.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
    ref.PackedUpdate.argtypes = [C.c_void_p, C.c_float, C.c_uint32, C.c_void_p]
    ref.FreePackArray.argtypes = [C.c_void_p]
    pack, plen = C.c_void_p(), C.c_uint32()
    ref.AllocPackArray(C.byref(pack), C.byref(plen), mass_len)
    ref.PackParticles(mass_len, part.ctypes.data, pack)
    scratch = part[:recv].copy()
    base = scratch.ctypes.data

    def work(t):
        for i in range(t, recv, cores):
            ref.PackedUpdate(base + 32 * i, DT, plen.value, pack)  # releases the GIL

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))
    sec = time.perf_counter() - t0
    ref.FreePackArray(pack)
    return sec


def pmc_traffic():
    """HBM bytes per step-kernel launch from the committed rocprofv3 PMC summary, if there is one."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", dest="n", type=int, default=N_PARTICLES,
                    help="particles (default 2^20, the size the metric is quoted on); not --n: torchrun claims that prefix")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--all-massive", action="store_true",
                    help="informational N^2 run: every particle is a source (not the BASELINE.json workload)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the multi-rank control flow (rendezvous, id broadcast, barriers, reduction, JSON) "
                         "without touching a GPU: no step runs and the reported value is 0")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import nbody_amd as nb  # loads libnbody_hip.so; aborts later if no gfx950 answers

    # torch.distributed only when launched through torch.distributed.run (also at world == 1, so that a
    # single-GPU box can rehearse the whole multi-rank flow with NB_HIP_FORCE_SHARDED=1)
    dist = None
    sharded = world > 1 or os.environ.get("NB_HIP_FORCE_SHARDED", "0") not in ("", "0")
    if "RANK" in os.environ and "MASTER_PORT" in os.environ:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    elif world > 1:
        sys.exit("WORLD_SIZE > 1 without a torch.distributed.run rendezvous (RANK / MASTER_PORT missing)")

    def barrier():
        if dist is not None:
            dist.barrier()

    if not args.dry_run:
        ndev = nb.device_count()
        nb.hip_lib().nb_hip_set_device(local_rank if local_rank < max(ndev, 1) else local_rank % max(ndev, 1))
    part, mass_len = make_workload(args.n, args.all_massive)
    n = part.shape[0]

    uid = None
    if sharded:
        # rank 0 makes the RCCL unique id; gloo carries its 128 bytes to the other ranks
        raw = bytearray(nb.comm_unique_id()) if rank == 0 else bytearray(nb.UNIQUE_ID_BYTES)
        if dist is not None:
            import torch

            buf = torch.frombuffer(raw, dtype=torch.uint8).clone()
            dist.broadcast(buf, src=0)
            raw = bytearray(buf.numpy().tobytes())
        uid = bytes(raw)

    if args.dry_run:
        assert uid is None or len(uid) == nb.UNIQUE_ID_BYTES
        plan = nb.shard_plan(n, mass_len, rank, world)
        assert plan["mass_count"] + plan["zero_count"] > 0 or n < world
        barrier()
        t0 = time.perf_counter()
        barrier()
        elapsed = max(time.perf_counter() - t0, 1e-9)
        kernel_ms, launches = 0.0, 0
        shape, info, sim = nb.plan_launch(plan["mass_count"] + plan["zero_count"], plan["src_padded"]), "dry-run", None
    else:
        sim = nb.SimPipeline(n, mass_len, rank=rank, nranks=world, unique_id=uid if sharded else None)
        if not sharded:
            sim.configure(graph=1)   # the K-step chain runs as a hipGraph on its first use, built inside the timed call
        sim.set_data(part)           # H2D + SoA split: outside the timed region

        if args.warmup > 0:
            sim.update(args.warmup, DT)
        barrier()
        sim.sync()
        t0 = time.perf_counter()
        sim.update(args.steps, DT)   # ONE call, K steps, blocking (hipGraph chain / RCCL-stepped chain)
        sim.sync()
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms, launches = sim.last_step_ms()

    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if sim is not None:
        shape = sim.launch_shape()
        info = nb.device_info()
        sim.close()

    if rank == 0:
        interactions = float(n) * float(mass_len) * args.steps
        value = 0.0 if args.dry_run else interactions / elapsed
        # dominant kernel: the step kernel; algorithmic flops per launch = interactions per launch * 14
        per_launch_s = (kernel_ms * 1e-3) / max(launches, 1)
        launch_interactions = float(n) * float(mass_len) / world * (args.steps / max(launches, 1))
        achieved_tflops = launch_interactions * FLOP_PER_INTERACTION / per_launch_s / 1e12 if per_launch_s > 0 else 0.0
        out = {
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
                "parallelism": f"receivers sharded N/{world} per GPU, all-gather of source positions per step"
                               if world > 1 else "single GPU",
                "kernel": shape,
                "device": info,
            },
            "roofline": {
                "bound": "valu",  # fp32 vector ALU (rsq + fma); neither HBM nor MFMA bounds this path
                "achieved": achieved_tflops,
                "peak": PEAK_FP32_VECTOR_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved_tflops / PEAK_FP32_VECTOR_TFLOPS,
                "traffic": None if args.all_massive else pmc_traffic(),
                "flop_per_interaction": FLOP_PER_INTERACTION,
                "kernel_ms_per_launch": per_launch_s * 1e3,
                "launches": launches,
            },
        }
        if world == 1 and not args.no_cpu_baseline and not args.dry_run:
            out["cpu_baseline"] = cpu_baseline(part, mass_len)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
