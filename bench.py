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

N = 1: the CPU baseline runs FIRST and the GPU legs last and back to back (headline K steps, two repeats of the
same K steps for the run-to-run spread, the LDS-tile route for roofline.alt_lds), so the GPU is busy for one
contiguous stretch an outside sampler can see.

N > 1: strong scaling -- the same 2^20 particles, N/P receivers per GPU, all-gather of source positions
per step over RCCL inside the library; torch.distributed (gloo) only carries the rendezvous, the barrier
and the reductions of the timings.  The line then also carries: what the RCCL communicator itself reports
(`rccl`), per-step kernel and all-gather time (`comm_ms_per_step`, `kernel_ms_per_step` min/max over ranks), a
self-check that all ranks hold the same state and that it matches a single-GPU run of the same steps
(`self_check`), and `extra_configs`: the overlapped step at the same N and BASELINE.json's config 5
(N = 2^22, plain and overlapped) from the same command.

Runtime note: under torch.distributed.run torch is imported before libnbody_hip.so is loaded, so the HIP runtime
and librccl that the data path binds are the ones bundled with the torch wheel (ROCm 7.0 build); a plain
`python bench.py` binds /opt/rocm's.  `runtime` in the JSON line says which.

The oracle (oracle/) is used here ONLY for the cpu_baseline leg.
"""
import argparse
import ctypes as C
import datetime
import hashlib
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
N_CONFIG5 = 1 << 22
DT = 0.01
KERNEL_SOURCES = ("nbody_amd/csrc/kernels.hip", "nbody_amd/csrc/kernels.h")


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


def kernel_sources_sha():
    """sha256 over the kernel sources with comments and blank space removed (editing a comment must not orphan a
    profile); with the launch shape and the source passes per step (both decided in pipeline.hip, both recorded next
    to the figure) it is what a committed PMC traffic figure is tied to."""
    import re

    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "r") as f:
            text = f.read()
        text = re.sub(r"// NB_HASH_OFF.*?// NB_HASH_ON[^\n]*", "", text, flags=re.S)   # host-side cost model (which shape
        #                                     it picks is recorded separately, under "launch")
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)     # block comments
        text = re.sub(r"//[^\n]*", "", text)                   # line comments (no string in these files holds "//")
        text = "\n".join(line.strip() for line in text.splitlines() if line.strip())
        h.update(text.encode())
    return h.hexdigest()


def pmc_traffic(n, shape=None, passes=None):
    """(HBM bytes per step-kernel launch, note): rocprofv3 PMC passes cannot run inside this process, so the figure comes
    from the committed profile -- and only counts while the kernel sources still hash to what was profiled and this run
    launched the same shape with the same number of source passes."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(p):
        return None, "no committed PMC profile"
    with open(p) as f:
        rec = json.load(f)
    if rec.get("n") not in (None, n):
        return None, f"committed PMC profile is for N={rec.get('n')}"
    want = rec.get("launch")
    if want is not None and shape is not None:
        got = dict(shape, passes=passes)
        if any(got.get(key) != val for key, val in want.items()):
            return None, f"stale: this run launched {got}, the PMC profile {rec.get('source')} was taken with {want}"
    if rec.get("kernel_sources_sha256") != kernel_sources_sha():
        return None, ("stale: kernel sources changed since the PMC profile " + str(rec.get("source"))
                      + " was taken (tools/profile.sh + tools/summarize_profile.py refresh it)")
    return rec.get("hbm_bytes_per_launch"), f"from {rec.get('source')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, same sources)"


def algorithmic_bytes_per_launch(n, m, passes):
    """DESIGN.md section 3 'Algorithmic bytes': per step, reads N*(pos 8 + radius 4) per pass + acc 8 per chained pass
    + vel 8 + M*(x, y, G*m) 12; writes acc 8 per pass + vel 8 + pos 8.  Mean per launch (= per pass)."""
    reads = n * (12 * passes + 8 * (passes - 1) + 8) + m * 12
    writes = n * (8 * passes + 16)
    return (reads + writes) / passes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", dest="n", type=int, default=N_PARTICLES,
                    help="particles (default 2^20, the size the metric is quoted on); not --n: torchrun claims that prefix")
    ap.add_argument("--extra-particles", dest="n5", type=int, default=N_CONFIG5,
                    help="size of the second sharded workload under extra_configs (default 2^22 = BASELINE.json config 5)")
    ap.add_argument("--transport", choices=("rccl", "host"), default="rccl",
                    help="N > 1: rccl = in-stream ncclAllGather (the product path); host = the library's caller-supplied "
                         "transport over torch.distributed gloo (host-staged, slow): lets several ranks share ONE GPU to "
                         "rehearse the multi-process flow where RCCL refuses duplicate devices")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline leg only: no repeats / alt_lds (1 GPU), no self_check / extra_configs (N GPUs)")
    ap.add_argument("--all-massive", action="store_true",
                    help="informational N^2 run: every particle is a source (not the BASELINE.json workload)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the multi-rank control flow (rendezvous, id broadcast, barriers, reductions, JSON keys) "
                         "without touching a GPU: no step runs and the reported value is 0")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

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

    # torch.distributed only when launched through torch.distributed.run (also at world == 1, so that a
    # single-GPU box can rehearse the whole multi-rank flow with NB_HIP_FORCE_SHARDED=1)
    dist = None
    torch = None
    if "RANK" in os.environ and "MASTER_PORT" in os.environ:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=900))
    elif world > 1:
        sys.exit("WORLD_SIZE > 1 without a torch.distributed.run rendezvous (RANK / MASTER_PORT missing)")

    def barrier():
        if dist is not None:
            dist.barrier()

    def reduce(values, op):
        """Element-wise MIN/MAX/SUM of a list of floats over the ranks."""
        if dist is None:
            return list(values)
        t = torch.tensor(list(values), dtype=torch.float64)
        dist.all_reduce(t, op={"max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN, "sum": dist.ReduceOp.SUM}[op])
        return [float(x) for x in t]

    def new_unique_id():
        """rank 0 makes an RCCL unique id; gloo carries its 128 bytes to the other ranks."""
        raw = bytearray(nb.comm_unique_id()) if rank == 0 else bytearray(nb.UNIQUE_ID_BYTES)
        if dist is not None:
            buf = torch.frombuffer(raw, dtype=torch.uint8).clone()
            dist.broadcast(buf, src=0)
            raw = bytearray(buf.numpy().tobytes())
        assert len(raw) == nb.UNIQUE_ID_BYTES
        return bytes(raw)

    gloo_gather = None
    if args.transport == "host" and sharded:
        def gloo_gather(rows, r, n):
            """In-place all-gather of host rows over gloo (rows[r] is filled on entry)."""
            if dist is None:
                return
            mine = torch.from_numpy(rows[r].copy())
            parts = [torch.empty_like(mine) for _ in range(n)]
            dist.all_gather(parts, mine)
            for q in range(n):
                if q != r:
                    rows[q] = parts[q].numpy()

    def make_sim(n_, m_):
        if not sharded:
            return nb.SimPipeline(n_, m_)
        if gloo_gather is not None:
            return nb.SimPipeline(n_, m_, rank=rank, nranks=world, allgather=gloo_gather)
        return nb.SimPipeline(n_, m_, rank=rank, nranks=world, unique_id=new_unique_id())

    if not args.dry_run:
        ndev = nb.device_count()
        nb.hip_lib().nb_hip_set_device(local_rank if local_rank < max(ndev, 1) else local_rank % max(ndev, 1))
    part, mass_len = make_workload(args.n, args.all_massive)
    n = part.shape[0]

    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and not args.dry_run:
        cpu = cpu_baseline(part, mass_len)   # before the GPU legs: those then run back to back

    def timed_leg(sim, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + device sync on both sides; max over ranks."""
        if warmup > 0:
            sim.update(warmup, DT)
        barrier()
        sim.sync()
        t0 = time.perf_counter()
        sim.update(steps, DT)   # ONE call, K steps, blocking (hipGraph chain / RCCL-stepped chain)
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
        lo = reduce([info["nranks"], info["rank"], info["device"]], "min")
        hi = reduce([info["nranks"], info["rank"], info["device"]], "max")
        sm = reduce([info["rank"], 1.0 if info["owns_comm"] else 0.0], "sum")
        return {
            "nranks_reported": {"min": int(lo[0]), "max": int(hi[0])},       # ncclCommCount on every rank
            "user_ranks": {"min": int(lo[1]), "max": int(hi[1]), "sum": int(sm[0])},  # ncclCommUserRank: 0..P-1, sum P(P-1)/2
            "devices": {"min": int(lo[2]), "max": int(hi[2])},              # ncclCommCuDevice
            "ranks_with_communicator": int(sm[1]),
            "version": info["rccl_version"],
            "lib": info["rccl_lib"],
            "first_gather_ms_rank0": info["first_gather_ms"],
        }

    extras = {}
    steps_done = 0
    if args.dry_run:
        uid = new_unique_id() if sharded else None
        assert uid is None or len(uid) == nb.UNIQUE_ID_BYTES
        plan = nb.shard_plan(n, mass_len, rank, world)
        assert plan["mass_count"] + plan["zero_count"] > 0 or n < world
        barrier()
        t0 = time.perf_counter()
        barrier()
        elapsed = reduce([max(time.perf_counter() - t0, 1e-9)], "max")[0]
        kernel_ms, launches, finish_launches = 0.0, 0, 0
        shape, info = nb.plan_launch(plan["mass_count"] + plan["zero_count"], plan["src_padded"]), "dry-run"
        if sharded:
            extras["rccl"] = {"nranks_reported": {"min": None, "max": None}, "user_ranks": None, "devices": None,
                              "ranks_with_communicator": 0, "version": None, "lib": None, "first_gather_ms_rank0": None}
            zero = {"min": 0.0, "max": 0.0}
            extras["comm_ms_per_step"], extras["kernel_ms_per_step"] = dict(zero), dict(zero)
            if not args.no_extras:
                digest = hashlib.sha256(part.tobytes()).digest()
                extras["self_check"] = {"ranks_agree": _digests_agree(dist, torch, digest), "vs_single_gpu_rel_l2_pos": None}
                part5, m5 = make_workload(args.n5)
                _ = new_unique_id()
                p5 = nb.shard_plan(part5.shape[0], m5, rank, world)
                assert p5["src_padded"] >= m5
                extras["extra_configs"] = [
                    _extra_entry(n, mass_len, 1, args.steps, 0.0, None, world),
                    _extra_entry(part5.shape[0], m5, 0, 3, 0.0, None, world),
                    _extra_entry(part5.shape[0], m5, 1, 3, 0.0, None, world),
                ]
        runtime = None
    else:
        sim = make_sim(n, mass_len)
        if not sharded:
            sim.configure(graph=1)   # the K-step chain runs as a hipGraph on its first use, built inside the timed call
        sim.set_data(part)           # H2D + SoA split: outside the timed region
        elapsed = timed_leg(sim, args.steps, args.warmup)
        steps_done = args.warmup + args.steps
        kernel_ms, launches = sim.last_step_ms()
        finish_launches = sim.finish_launches()
        shape = sim.launch_shape()
        info = nb.device_info()
        runtime = {"hip_runtime_version": int(nb.hip_lib().nb_hip_runtime_version()),
                   "torch_imported_first": torch is not None}

        if sharded:
            extras["rccl"] = comm_evidence(sim)
            d = sharded_detail(sim, args.steps)
            extras["comm_ms_per_step"], extras["kernel_ms_per_step"] = d["comm_ms_per_step"], d["kernel_ms_per_step"]
            if not args.no_extras:
                # every rank must hold the same full state, and it must be the single-GPU state of the same steps
                got = sim.get_data()  # collective
                check = {"ranks_agree": _digests_agree(dist, torch, hashlib.sha256(got.tobytes()).digest()),
                         "steps": steps_done}
                if rank == 0:
                    one = nb.SimPipeline(n, mass_len)
                    one.set_data(part)
                    one.update(steps_done, DT)
                    want = one.get_data()
                    one.close()
                    dp = (got[:, 0:2].astype(np.float64) - want[:, 0:2]).ravel()
                    check["vs_single_gpu_rel_l2_pos"] = float(np.sqrt(dp @ dp) / np.linalg.norm(want[:, 0:2].astype(np.float64)))
                    check["vs_single_gpu_max_abs_pos"] = float(np.abs(dp).max())
                    check["static_fields_equal"] = bool(np.array_equal(got[:, 6:8], want[:, 6:8]))
                barrier()
                extras["self_check"] = check
                # the overlapped step on the same pipeline
                sim.configure(overlap=1)
                e1 = timed_leg(sim, args.steps, 1)
                extra = [_extra_entry(n, mass_len, 1, args.steps, e1, sharded_detail(sim, args.steps), world)]
                sim.close()
                sim = None
                # BASELINE.json config 5: N = 2^22, plain and overlapped (own communicator: a second ncclCommInitRank)
                part5, m5 = make_workload(args.n5)
                sim5 = make_sim(part5.shape[0], m5)
                sim5.set_data(part5)
                for ov in (0, 1):
                    sim5.configure(overlap=ov)
                    e5 = timed_leg(sim5, 3, 1)
                    extra.append(_extra_entry(part5.shape[0], m5, ov, 3, e5, sharded_detail(sim5, 3), world))
                sim5.close()
                extras["extra_configs"] = extra
        elif not args.no_extras:
            # same K steps twice more (run-to-run spread), then the LDS-tile route of the north star on the same chain
            extras["repeat_ms_per_step"] = [timed_leg(sim, args.steps, 0) / args.steps * 1e3 for _ in range(2)]
            sim.configure(variant=0)
            e_lds = timed_leg(sim, args.steps, 2)
            lds_ms, lds_launches = sim.last_step_ms()
            extras["alt_lds"] = (e_lds, lds_ms, lds_launches, sim.launch_shape())
        if sim is not None:
            sim.close()

    if rank == 0:
        interactions = float(n) * float(mass_len) * args.steps
        value = 0.0 if args.dry_run else interactions / elapsed
        # dominant kernel: the step kernel; algorithmic flops per launch = interactions per launch * 14
        per_launch_s = (kernel_ms * 1e-3) / max(launches, 1)
        launch_interactions = float(n) * float(mass_len) / world * (args.steps / max(launches, 1))
        achieved_tflops = launch_interactions * FLOP_PER_INTERACTION / per_launch_s / 1e12 if per_launch_s > 0 else 0.0
        passes = max(launches // max(args.steps, 1), 1)
        traffic, traffic_note = (None, "not measured for this workload") if (args.all_massive or world > 1 or args.dry_run) \
            else pmc_traffic(n, shape, passes)
        roof = {
            "bound": "valu",  # fp32 vector ALU (rsq + fma); neither HBM nor MFMA bounds this path
            "achieved": achieved_tflops,
            "peak": PEAK_FP32_VECTOR_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved_tflops / PEAK_FP32_VECTOR_TFLOPS,
            "traffic": traffic,
            "traffic_note": traffic_note,
            "traffic_algorithmic": algorithmic_bytes_per_launch(n, mass_len, passes) if world == 1 else None,
            "flop_per_interaction": FLOP_PER_INTERACTION,
            "kernel_ms_per_launch": per_launch_s * 1e3,
            "launches": launches,
            "finish_launches": finish_launches,
            "kernel_ms_note": ("HIP events on the launch stream around the whole chain / step-kernel launches"
                               + ("; each interval also holds one O(N) finish kernel (~9 us at 2^20) per step launch"
                                  if finish_launches else "")
                               + ("; sharded: the interval includes the all-gathers, see kernel_ms_per_step" if sharded else "")),
        }
        if "alt_lds" in extras:
            e_lds, lds_ms, lds_launches, lds_shape = extras.pop("alt_lds")
            lds_s = lds_ms * 1e-3 / max(lds_launches, 1)
            lds_tf = float(n) * float(mass_len) * (args.steps / max(lds_launches, 1)) * FLOP_PER_INTERACTION / lds_s / 1e12
            roof["alt_lds"] = {"note": "same K steps through the LDS-tile source route (north star's design; variant=0), "
                                       "bit-identical results, timed after the headline leg",
                               "ms_per_step": e_lds / args.steps * 1e3, "value": interactions / e_lds,
                               "kernel_ms_per_launch": lds_s * 1e3, "achieved": lds_tf,
                               "frac": lds_tf / PEAK_FP32_VECTOR_TFLOPS, "kernel": lds_shape}
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
                "parallelism": (f"receivers sharded N/{world} per GPU, all-gather of source positions per step"
                                + (" over the caller-supplied HOST transport (gloo; rehearsal, not RCCL)" if gloo_gather else ""))
                               if world > 1 else "single GPU",
                "kernel": shape,
                "device": info,
            },
            "roofline": roof,
            "runtime": runtime,
        }
        if sharded:
            # ncclCommCount as seen by every rank's communicator -- null when a rank holds none (host transport, dry run)
            out["rccl_nranks"] = (extras["rccl"]["nranks_reported"]["min"]
                                  if extras["rccl"]["ranks_with_communicator"] == world else None)
            out["transport"] = "host (gloo all-gather through page-locked staging)" if gloo_gather else "rccl (in-stream ncclAllGather)"
        out.update(extras)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _digests_agree(dist, torch, digest):
    """True when every rank's sha256 equals rank 0's."""
    if dist is None:
        return True
    mine = torch.frombuffer(bytearray(digest), dtype=torch.uint8).clone()
    ref = mine.clone()
    dist.broadcast(ref, src=0)
    same = torch.tensor([1.0 if bool((mine == ref).all()) else 0.0], dtype=torch.float64)
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    return bool(same.item() == 1.0)


def _extra_entry(n, m, overlap, steps, elapsed, detail, world):
    e = {
        "workload": f"srand(11037) MakeGalaxies({n}, 2), N={n}, mass_len={m}, dt={DT}, N/{world} receivers per GPU",
        "overlap": overlap,   # 1 = own-shard kernel runs while the other shards' positions are still being gathered
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


if __name__ == "__main__":
    main()
