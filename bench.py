#!/usr/bin/env python3
"""bench.py -- the hot path's headline number: particle-pair interactions/s at N = 2^20 on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one process per GPU
through torch.distributed.run.  One JSON line on rank 0.  `python bench.py --gpus N` started BARE works too: the process
that is started never touches the GPU -- it is a supervisor over fresh rank processes (supervise()),
which is also what every rank process is under torch.distributed.run.

Workload (BASELINE.json metric): srand(11037); MakeGalaxies(2^20, 2) -- the reference bench's universe
(src/bench.c:42,53) at the size the metric is quoted on -- partitioned by CreateWorld, dt = 0.01.  A "step" is one force +
integrate pass over all N receivers against all mass_len sources.  Interactions per step = N * mass_len (what the
reference kernels evaluate, particle_cs.glsl:30,35).  K steps run as ONE PerformSimUpdate(K) call, like the reference
harness' update(w, dt, 100) (bench.c:30-33); particles are resident in HBM before the timed region (SetSimulationData is
outside it).

N = 1: the CPU baseline runs FIRST (the reference's own UpdateWorld_CPU, compiled where it lies, and the product's on the
same World and thread count), then the GPU legs back to back: headline K steps -- from here on the line is in hand and
every later leg runs under a C-level last-gasp handler --, the clock probe, the headline's parity stamp, two repeats of
the same K steps (run-to-run spread), the same K steps with the clock sampler beside them (roofline.held_clock_ghz,
cycles_per_wave_interaction), the LDS-tile route (roofline.alt_lds), and `extra_configs`: BASELINE.json's other
single-GPU configurations in the reference harness' shape -- C2 (N = 65 536, one 100-step call after 10 warm-up steps),
C3 (N = 262 144, a cached hipGraph chain at dt, then the same chain at dt/2), N2 (all-massive), C1 (N = 4 096, the
product's nbody-bench --cpu, no GPU) -- and the per-rank shard steps S2 / S4 / S8 / C5S8: what ONE rank's step of a 2 / 4 /
8-way sharded run costs, measured on this one GPU (the compute half of the 1/2/4/8 curve).

N > 1: strong scaling -- the same 2^20 particles, N/P receivers per GPU, all-gather of source positions per step over RCCL
inside the library.  `--transport auto` (default): an RCCL attempt whose ranks do not all deliver a complete headline is
followed by a FRESH set of rank processes over the direct device-to-device exchange; the line then carries
"transport_fallback".  The rendezvous, the barriers and the reductions of the timings go over a stdlib Unix-socket hub
between the ranks (nbody_amd/ranklink.py): torch is NOT imported, so the HIP runtime and the librccl the data path binds
are /opt/rocm's -- the stack every single-GPU test runs on (`--rendezvous gloo` keeps the round-3 route).  The JSON dict
is COMPLETE after the headline leg and the self-check (mandatory for every N > 1 headline: all ranks agree and match a
single-GPU run; what the RCCL communicator itself reports; per-step kernel / all-gather times); every later leg
(`extra_configs`: overlapped step, the chain captured as a hipGraph, BASELINE.json's config 5 at N = 2^22 plain and
overlapped, the direct exchange with its own self-check) runs under a host-side deadline: if one stalls, rank 0 writes
the line with what is in hand plus "extras_aborted": "<leg>" and every rank leaves with exit code 4 -- a fresh exit,
never a re-exec.

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

FLOP_PER_INTERACTION = 14        # reference op count, sim_cpu.c:169-188
PEAK_FP32_VECTOR_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (vector)"
N_PARTICLES = 1 << 20
N_CONFIG5 = 1 << 22
N_CONFIG2 = 1 << 16
N_CONFIG3 = 1 << 18
DT = 0.01
KERNEL_SOURCES = ("nbody_amd/csrc/kernels.hip", "nbody_amd/csrc/kernels.h", "nbody_amd/csrc/interaction_asm.h")


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

def host_cpu_share():
    """How many host threads the CPU legs may use, and why.  Derived from what the process is allowed to run on -- the
    scheduler affinity mask and, where the container sets one, the cgroup CPU quota -- not from a literal.  Only when
    neither narrows a big host (affinity == every core of a > 32-core machine, no quota) does the pool's documented share
    apply (one GPU of this pool comes with 16 CPUs); NB_BENCH_CPU_THREADS overrides everything.  Everything consulted is
    recorded on the line."""
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    count = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:      # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
                if q > 0 and per > 0:
                    quota = max(1, int(q / per + 0.5))
        except (OSError, ValueError):
            pass
    threads, why = (affinity or count), "sched_getaffinity"
    if quota is not None and quota < threads:
        threads, why = quota, "cgroup cpu quota"
    if why == "sched_getaffinity" and threads == count and count > 32:
        threads, why = 16, "pool share (16 CPUs per GPU box; affinity and cgroup quota leave all %d cores open)" % count
    env = os.environ.get("NB_BENCH_CPU_THREADS")
    if env and env.isdigit() and int(env) > 0:
        threads, why = int(env), "NB_BENCH_CPU_THREADS"
    return {"threads": max(1, threads), "threads_from": why, "affinity_cores": affinity, "os_cpu_count": count,
            "cgroup_cpu_quota": quota, "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS")}


def host_cores():
    return host_cpu_share()["threads"]


def _libgomp():
    for name in ("libgomp.so.1", "libgomp.so"):
        try:
            return C.CDLL(name)
        except OSError:
            pass
    return None


def cpu_baseline(part, mass_len, budget_s=12.0):
    """The reference's own UpdateWorld_CPU (src/lib/world.c:99-110: PackParticles + the OpenMP schedule(static, 20) loop
    over PackedUpdate, src/lib/sim_cpu.c:156-194), compiled where it lies into oracle/_ref/libnbody_ref_world.so, timed
    on this box's host cores over a bounded sample of the same workload; beside it the product's UpdateWorld_CPU on the
    same World with the same number of threads.  Fallbacks, in order, when that library is absent: the
    reference's PackedUpdate object code driven per receiver from Python threads (oracle/_ref/libnbody_ref_cpu.so), then
    the oracle's AVX restatement (kind "port")."""
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
    if sec is None and os.path.exists(ob.REF_CPU_SO):
        try:
            sec = _time_reference_packedupdate(ob.REF_CPU_SO, part, mass_len, recv, cores)
            kind = "reference"
            how = "reference PackedUpdate object code (oracle/_ref/libnbody_ref_cpu.so) driven per receiver from Python threads"
        except Exception as e:  # pragma: no cover - diagnostic only
            print(f"[bench] reference PackedUpdate leg failed ({e}); using the port", file=sys.stderr)
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
        "cpu_model": _cpu_model(),
        "value_1_thread": one,
        "sample": f"World of the first {recv} of {n} partitioned particles (all {mass_len} sources), one step, AVX (-mavx, no FMA) "
                  f"+ {threads} threads ({sec:.2f} s); the whole step would take ~{sec * n / recv:.0f} s",
    }
    out.update({k: share[k] for k in ("threads_from", "affinity_cores", "os_cpu_count", "cgroup_cpu_quota", "OMP_NUM_THREADS")})
    try:
        out["product"] = product_cpu_leg(part, mass_len, recv, threads)
    except Exception as e:  # pragma: no cover - diagnostic only
        out["product"] = {"error": str(e)}
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
    gomp = _libgomp()
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
    gomp = _libgomp()
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


# ---- parity stamp ----------------------------------------------------------------------------------------------------

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


# ---- roofline.traffic: tied to the committed PMC profile ----------------------------------------------------------------

def kernel_sources_sha():
    """sha256 over the kernel sources with comments and blank space removed (editing a comment must not orphan a
    profile); with the launch shape and the source passes per step (both decided in step_chain.hip, both recorded next
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


def _pmc_record(n, shape=None, passes=None):
    """(record, note): the committed PMC profile, or None and why it does not apply to this run."""
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
    return rec, f"from {rec.get('source')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, same sources)"


def pmc_traffic(n, shape=None, passes=None):
    """(HBM bytes per step-kernel launch, note): rocprofv3 PMC passes cannot run inside this process, so the figure comes
    from the committed profile -- and only counts while the kernel sources still hash to what was profiled and this run
    launched the same shape with the same number of source passes."""
    rec, note = _pmc_record(n, shape, passes)
    return (rec.get("hbm_bytes_per_launch") if rec else None), note


def pmc_traffic_parts(n, shape=None, passes=None):
    """The same figure taken apart: FETCH_SIZE as counted (raw), with the guide's x2, and WRITE_SIZE -- so a reader can
    see which part of `traffic` is a measurement and which a correction."""
    rec, _ = _pmc_record(n, shape, passes)
    if not rec or "fetch_bytes_raw" not in rec:
        return None
    return {"fetch_raw": rec["fetch_bytes_raw"], "fetch_corrected_x2": 2.0 * rec["fetch_bytes_raw"], "write": rec["write_bytes"],
            "raw_total": rec["fetch_bytes_raw"] + rec["write_bytes"],
            "note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE.  MI355X_MICROARCH.md calibrates the x2 on 16-B-per-lane "
                    "coalesced streaming loads; this kernel's global loads are 8-B float2 / 4-B float per lane plus the "
                    "scalar cache's 64-B line fills (uncalibrated widths), so the corrected figure is an upper bound and "
                    "raw_total a lower bound.  Either way ~1-2 GB/s of ~8000: HBM does not bound this kernel."}


def algorithmic_bytes_per_launch(n, m, passes):
    """DESIGN.md section 3 'Algorithmic bytes': per step, reads N*(pos 8 + radius 4) per pass + acc 8 per chained pass
    + vel 8 + M*(x, y, G*m) 12; writes acc 8 per pass + vel 8 + pos 8.  Mean per launch (= per pass)."""
    reads = n * (12 * passes + 8 * (passes - 1) + 8) + m * 12
    writes = n * (8 * passes + 16)
    return (reads + writes) / passes


MIX_FLOOR_CYCLES = 26.0   # 9 plain fp32 VALU instructions at 2 issue cycles + one v_rsq_f32 at 8
NOMINAL_CLOCK_GHZ = 2.4   # MI355X_MICROARCH.md "Max clock"; the 157.3 TFLOP/s peak is quoted at it


def device_cus(device_info):
    """Compute units out of nb_hip_device_info's "name arch CUs clockMHz pci=..." (the token after the gfx arch)."""
    tok = str(device_info).split()
    for i, t in enumerate(tok):
        if t.startswith("gfx") and i + 1 < len(tok) and tok[i + 1].isdigit():
            return int(tok[i + 1])
    return 256


def held_clock_fields(probe, sampled, per_launch_s, launch_interactions, device_info, achieved_tflops):
    """roofline.held_clock_ghz and what follows from it: how many shader cycles one wave-interaction of the TIMED kernel
    took on every SIMD (kernel seconds x held clock x SIMDs / wave-interactions), which fraction of the instruction mix's
    26-cycle floor that is, and the roofline fraction re-priced at the held clock instead of the nominal 2.4 GHz.
    `sampled` (preferred): the clock sampler's reading during a repeat of the same K steps -- the clock the chip holds
    under the step kernel itself.  `probe`: the separate probe kernel run right after the headline leg; its loop is denser
    than the step kernel's, so the chip holds a lower clock for it -- kept on the line as the pure-loop reference."""
    out = {"held_clock_ghz": None, "clock_probe": probe, "clock_sampled": sampled}
    ghz, source, slowest = None, None, None
    if sampled and sampled.get("clock_ghz"):
        per_xcd = [v for v in sampled.get("per_xcd_ghz", []) if v > 0]
        # every XCD computes an eighth of a launch (the dispatcher deals workgroups round-robin to the XCDs), so the chip's
        # clock is the mean over the XCDs -- and the launch ends with its slowest XCD
        ghz = sum(per_xcd) / len(per_xcd) if per_xcd else sampled["clock_ghz"]
        slowest = min(per_xcd) if per_xcd else None
        source = ("clock sampler during a repeat of the same K steps (8 one-wave workgroups, one per XCD, stamping s_memtime / "
                  "s_memrealtime every 0.5 ms on their own stream, outside the headline's timed region); mean of the per-XCD medians")
    elif probe and probe.get("clock_ghz"):
        ghz, source = probe["clock_ghz"], "probe kernel right after the headline leg (reads LOW: its loop is denser than the step kernel's)"
    if ghz is None or per_launch_s <= 0 or launch_interactions <= 0:
        return out
    simds = 4 * device_cus(device_info)
    cycles = per_launch_s * ghz * 1e9 * simds / (launch_interactions / 64.0)
    out.update({
        "held_clock_ghz": ghz,
        "held_clock_source": source,
        "cycles_per_wave_interaction": cycles,
        "frac_of_mix_ceiling": MIX_FLOOR_CYCLES / cycles,
        "held_clock_ghz_slowest_xcd": slowest,
        # workgroups are dealt to the XCDs in equal shares, so a launch lasts as long as its slowest XCD needs
        "cycles_per_wave_interaction_slowest_xcd": cycles * slowest / ghz if slowest else None,
        "frac_at_held_clock": achieved_tflops / (PEAK_FP32_VECTOR_TFLOPS * ghz / NOMINAL_CLOCK_GHZ),
        # one wave-interaction = 14 x 64 counted flops; the peak is 64 flop per cycle and SIMD (157.3e12 / 1024 / 2.4e9)
        "mix_ceiling_frac_at_nominal_clock": FLOP_PER_INTERACTION / MIX_FLOOR_CYCLES,
        "cycles_note": f"cycles_per_wave_interaction = headline kernel seconds per launch x held clock x {simds} SIMDs / wave-interactions "
                       f"per launch; floor of this instruction mix = {MIX_FLOOR_CYCLES:g} cycles (9 plain fp32 VALU x 2 + v_rsq_f32 x 8)",
    })
    return out


# ---- deadline guard of the optional legs --------------------------------------------------------------------------------

class LastGasp:
    """The optional legs can also die the hard way: the library's error convention is the reference's -- print and abort()
    (src/lib/util.h:17-29) -- and RCCL inside stream capture with several ranks has never run anywhere.  A fatal signal
    raised inside a C call never reaches a Python-level handler, so rank 0 registers a C one while optional legs run
    (nbody_amd/csrc/last_gasp.c -> lib/libnb_lastgasp.so): on SIGABRT / SIGSEGV / SIGBUS / SIGFPE it write()s the line
    prepared when the current leg was armed -- headline + self-check + the legs finished so far + "extras_aborted" -- and
    _exit(6)s.  The handler is plain C doing only async-signal-safe calls on bytes copied beforehand: no GIL, no
    allocation, no Python in signal context."""

    def __init__(self, fd):
        self.fd = fd
        self._lib = C.CDLL(os.path.join(ROOT, "nbody_amd", "lib", "libnb_lastgasp.so"))
        self._lib.nb_last_gasp_set.argtypes = [C.c_int, C.c_char_p, C.c_ulong]
        self._lib.nb_last_gasp_set.restype = C.c_int

    def arm(self, line_bytes):
        if self._lib.nb_last_gasp_set(self.fd, line_bytes, len(line_bytes)) != 0:
            print("[bench] last-gasp line too long; keeping the previous one", file=sys.stderr)

    def disarm(self):
        self._lib.nb_last_gasp_disarm()


class LegGuard:
    """Host-side deadline around every optional leg.  The legs block inside C calls (ctypes releases the GIL), so a
    Python thread can watch the clock: on expiry rank 0 writes the JSON line with what is in hand plus
    "extras_aborted", and every rank leaves with os._exit(4) -- a fresh exit (the process has touched the GPU; no
    re-exec, no retry).  The other ranks wait a moment first so that rank 0's line is out before the launcher reacts."""

    def __init__(self, rank, emit_partial, default_s):
        self.rank, self.emit_partial, self.default_s = rank, emit_partial, default_s
        self.lock = threading.Lock()
        self.leg, self.until = None, None
        self.thread = threading.Thread(target=self._watch, daemon=True)
        self.thread.start()

    def arm(self, leg, seconds=None):
        with self.lock:
            self.leg, self.until = leg, time.monotonic() + (seconds if seconds else self.default_s)

    def disarm(self):
        with self.lock:
            self.leg, self.until = None, None

    def _watch(self):
        while True:
            time.sleep(0.2)
            with self.lock:
                leg, until = self.leg, self.until
            if leg is None or time.monotonic() < until:
                continue
            print(f"[bench] rank {self.rank}: leg '{leg}' passed its deadline; writing what is in hand and exiting (4)",
                  file=sys.stderr, flush=True)
            if self.rank == 0:
                self.emit_partial(leg)
            else:
                time.sleep(3.0)
            os._exit(4)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", dest="n", type=int, default=N_PARTICLES,
                    help="particles (default 2^20, the size the metric is quoted on); not --n: torchrun claims that prefix")
    ap.add_argument("--extra-particles", dest="n5", type=int, default=N_CONFIG5,
                    help="size of the second sharded workload under extra_configs (default 2^22 = BASELINE.json config 5)")
    ap.add_argument("--transport", choices=("auto", "rccl", "host", "direct"), default="auto",
                    help="N > 1: auto (default) = rccl, and if any rank of that attempt leaves before the headline is in hand, a FRESH "
                         "set of rank processes with the direct exchange (the line is stamped transport_fallback); "
                         "rccl = in-stream ncclAllGather (the product path); direct = no RCCL: every rank pushes its slice "
                         "device-to-device into its peers' IPC-mapped source arrays, one barrier per step over the rendezvous link "
                         "(the fallback should RCCL not come up); host = data staged through the host over the rendezvous link "
                         "(slow).  direct and host let several ranks share ONE GPU, where RCCL refuses duplicate devices")
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
                    help="rehearse the multi-rank control flow (rendezvous, id broadcast, barriers, reductions, JSON keys) "
                         "without touching a GPU: no step runs and the reported value is 0")
    ap.add_argument("--leg-deadline-s", type=float, default=120.0,
                    help="N > 1: host-side deadline of every optional leg (below the library's own 180 s collective watchdog)")
    ap.add_argument("--stall-leg", default=None,
                    help="rehearsal only (host transport): the all-gather callback never returns during this leg "
                         "(overlap | sharded_graph | config5), to exercise the deadline path")
    ap.add_argument("--crash-leg", default=None,
                    help="rehearsal only: rank 0 abort()s inside this leg (N > 1, host transport: in the leg's all-gather; one GPU: "
                         "at the start of the leg -- 'clock probe', 'parity stamp', 'repeats', 'clock sampler leg', 'alt_lds', "
                         "'extra_configs C2/C3/N2/C1', 'extra_configs S2/S4/S8/C5S8'), to exercise the last-gasp line")
    ap.add_argument("--rehearse-rccl-failure", action="store_true",
                    help="rehearsal only: in an rccl attempt the last rank leaves with exit code 3 right after the rendezvous, like the "
                         "library's watchdog does when ncclCommInitRank never completes -- exercises --transport auto's fallback")
    ap.add_argument("--rehearse-hang", action="store_true",
                    help="rehearsal only: every rank sleeps for a minute right after the rendezvous (a collective that never "
                         "completes), to exercise the supervisor's clean-up when it is told to stop")
    ap.add_argument("--attempt-timeout-s", type=float, default=900.0,
                    help="N > 1: the supervisor ends (by exact pid) rank processes of an attempt that runs longer than this")
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

    # The ranks meet on the host only when launched through torch.distributed.run (also at world == 1, so that a
    # single-GPU box can rehearse the whole multi-rank flow with NB_HIP_FORCE_SHARDED=1): by default over a stdlib
    # socket hub, so that no torch -- and with it no second HIP runtime / librccl -- is in the process.
    dist = None
    torch = None
    link = None
    if "RANK" in os.environ and "MASTER_PORT" in os.environ:
        if args.rendezvous == "gloo":
            import torch
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=900))
        else:
            from nbody_amd.ranklink import RankLink

            link = RankLink(rank, world, name="nbody_bench_%s_%s_a%d" % (os.environ["MASTER_PORT"],
                                                                        os.environ.get("TORCHELASTIC_RUN_ID", "none"), attempt))
            if args.rehearse_hang:
                link.barrier()
                time.sleep(60.0)
                os._exit(9)
            if args.rehearse_rccl_failure and args.transport == "rccl" and world > 1:
                link.barrier()
                if rank == world - 1:
                    print(f"[bench] rank {rank}: rehearsing an RCCL bootstrap that never completes: exit 3", file=sys.stderr, flush=True)
                    os._exit(3)
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

    def new_unique_id():
        """rank 0 makes an RCCL unique id; the rendezvous carries its 128 bytes to the other ranks."""
        if args.dry_run:   # no GPU: /opt/rocm's ncclGetUniqueId needs one; the hand-over is what is rehearsed
            raw = bytearray(range(nb.UNIQUE_ID_BYTES)) if rank == 0 else bytearray(nb.UNIQUE_ID_BYTES)
        else:
            raw = bytearray(nb.comm_unique_id()) if rank == 0 else bytearray(nb.UNIQUE_ID_BYTES)
        if link is not None:
            raw = bytearray(link.broadcast(bytes(raw), src=0))
        elif dist is not None:
            buf = torch.frombuffer(raw, dtype=torch.uint8).clone()
            dist.broadcast(buf, src=0)
            raw = bytearray(buf.numpy().tobytes())
        assert len(raw) == nb.UNIQUE_ID_BYTES and (not args.dry_run or bytes(raw) == bytes(range(nb.UNIQUE_ID_BYTES)))
        return bytes(raw)

    def digests_agree(digest):
        """True when every rank's sha256 equals rank 0's."""
        if link is not None:
            return all(d == digest for d in link.allgather(bytes(digest)))
        return _digests_agree(dist, torch, digest)

    current_leg = {"name": "headline"}
    host_gather = None
    link_gather = None
    if sharded:
        def link_gather(rows, r, n):
            """In-place all-gather of host rows over the rendezvous (rows[r] is filled on entry)."""
            if args.stall_leg and current_leg["name"] == args.stall_leg:
                time.sleep(3600.0)   # rehearsal of a collective that never completes (--stall-leg)
            if args.crash_leg and current_leg["name"] == args.crash_leg and r == 0:
                os.abort()           # rehearsal of a leg that dies by the library's abort() convention (--crash-leg)
            if link is not None:
                for q, row in enumerate(link.allgather(rows[r].tobytes())):
                    if q != r:
                        rows[q] = np.frombuffer(row, dtype=rows.dtype)
                return
            if dist is None:
                return
            mine = torch.from_numpy(rows[r].copy())
            parts = [torch.empty_like(mine) for _ in range(n)]
            dist.all_gather(parts, mine)
            for q in range(n):
                if q != r:
                    rows[q] = parts[q].numpy()

    if args.transport in ("host", "direct") and sharded:
        host_gather = link_gather

    def make_sim(n_, m_):
        if not sharded:
            return nb.SimPipeline(n_, m_)
        if host_gather is not None:
            return nb.SimPipeline(n_, m_, rank=rank, nranks=world, allgather=host_gather, direct=args.transport == "direct")
        return nb.SimPipeline(n_, m_, rank=rank, nranks=world, unique_id=new_unique_id())

    if not args.dry_run:
        ndev = nb.device_count()
        nb.hip_lib().nb_hip_set_device(local_rank if local_rank < max(ndev, 1) else local_rank % max(ndev, 1))
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

    # ---- the headline leg ---------------------------------------------------------------------------------------
    extras = {}
    steps_done = 0
    sim = None
    if args.dry_run:
        uid = new_unique_id() if sharded else None
        assert uid is None or len(uid) == nb.UNIQUE_ID_BYTES
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
                              "ranks_with_communicator": 0, "version": None, "lib": None, "first_gather_ms_rank0": None}
            zero = {"min": 0.0, "max": 0.0}
            extras["comm_ms_per_step"], extras["kernel_ms_per_step"] = dict(zero), dict(zero)
        runtime = None
    else:
        sim = make_sim(n, mass_len)
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

    # ---- the JSON dict: complete from here on; later legs only add keys -----------------------------------------
    out = {}
    out_lock = threading.Lock()
    emitted = {"done": False}

    def emit(extra_keys=None):
        """Write the one line (rank 0, once)."""
        with out_lock:
            if emitted["done"] or rank != 0:
                return
            emitted["done"] = True
            line = dict(out)
            if extra_keys:
                line.update(extra_keys)
            os.write(json_fd, (json.dumps(line) + "\n").encode())

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
            "traffic": traffic,
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
                                       " over the caller-supplied HOST transport (rehearsal, not RCCL)" if host_gather else ""))
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

    def put(key, val):
        if rank == 0:
            with out_lock:
                out[key] = val

    # Single GPU: from here on every further leg -- clock probe, parity stamp, repeats, the clock-sampler leg, the LDS
    # route, extra_configs -- runs with the line in hand: the library's error convention is abort(), and a fatal signal
    # inside any of them still writes the headline (plus "extras_aborted": which leg) through the C-level handler.
    solo_gasp = LastGasp(json_fd) if (rank == 0 and not sharded and not args.dry_run) else None

    def solo_leg(name):
        if solo_gasp:
            with out_lock:
                solo_gasp.arm((json.dumps(dict(out, extras_aborted=f"{name} (fatal signal)")) + "\n").encode())
            if args.crash_leg == name:
                os.abort()   # rehearsal of a leg that dies by the library's abort() convention (--crash-leg)

    if not sharded and not args.dry_run and rank == 0:
        if not args.no_clock_probe:
            # the clock the chip holds for the interaction statement alone, asked right after the timed steps and outside
            # them: a separate probe kernel (include/nbody_hip.h nb_hip_probe_clock); the product kernels carry no stamps
            solo_leg("clock probe")
            try:
                clock = nb.probe_clock(40.0)
            except Exception as e:  # pragma: no cover - diagnostic only
                clock = {"error": str(e)}
            with out_lock:
                out["roofline"].update(held_clock_fields(clock, None, per_launch_s, launch_interactions, info, achieved_tflops))
        if not args.no_parity:
            solo_leg("parity stamp")
            put("parity", parity_stamp(sim, mass_len))   # after the timed call, outside it

    # ---- optional legs ------------------------------------------------------------------------------------------
    if sharded:
        guard = LegGuard(rank, lambda leg: emit({"extras_aborted": leg}), args.leg_deadline_s) if not args.dry_run else None
        gasp = LastGasp(json_fd) if (rank == 0 and not args.dry_run) else None

        def leg(name):
            current_leg["name"] = name
            if guard:
                guard.arm(name)
            if gasp:
                with out_lock:
                    gasp.arm((json.dumps(dict(out, extras_aborted=f"{name} (fatal signal)")) + "\n").encode())

        if args.dry_run:
            digest = hashlib.sha256(part.tobytes()).digest()
            put("self_check", {"ranks_agree": digests_agree(digest), "vs_single_gpu_rel_l2_pos": None})
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
        else:
            # NOT optional for any N > 1 headline (--no-extras keeps it): every rank must hold the same full state, and it must
            # be the single-GPU state of the same steps -- the check that would catch a stale or torn exchange
            single_gpu_state = {}

            def self_check(pipeline, steps_run):
                got = pipeline.get_data()  # collective
                check = {"ranks_agree": digests_agree(hashlib.sha256(got.tobytes()).digest()), "steps": steps_run}
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
                pci = nb.device_info().split("pci=")[-1].encode()
                seen = sorted(set(link.allgather(pci))) if link is not None else [pci]
                check["devices"] = [d.decode() for d in seen]
                check["crossed_devices"] = len(seen) > 1
                barrier()
                return check

            leg("self_check")
            put("self_check", self_check(sim, steps_done))
            if args.no_extras:
                if guard:
                    guard.disarm()
                if gasp:
                    gasp.disarm()
                sim.close()
                sim = None
        if not args.dry_run and not args.no_extras:
            extra = []
            put("extra_configs", extra)   # the list grows in place: a deadline line carries the legs that finished
            # the overlapped step on the same pipeline
            leg("overlap")
            sim.configure(overlap=1)
            e1 = timed_leg(sim, args.steps, 1)
            extra.append(_extra_entry(n, mass_len, 1, 0, args.steps, e1, sharded_detail(sim, args.steps), world))
            sim.configure(overlap=0)
            # BASELINE.json config 5: N = 2^22, plain and overlapped (own communicator: a second ncclCommInitRank)
            leg("config5")
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
            if host_gather is None and link_gather is not None:
                leg("direct")
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
            leg("sharded_graph")
            if host_gather is None:
                sim.configure(sharded_graph=1)
                eg = timed_leg(sim, args.steps, args.steps)   # the warm-up call captures and instantiates the chain
                entry = _extra_entry(n, mass_len, 0, 1, args.steps, eg, None, world)
                entry["graph_stats"] = sim.graph_stats()
                extra.append(entry)
                sim.configure(sharded_graph=0)
            else:
                if args.stall_leg == "sharded_graph":
                    sim.update(1, DT)
                extra.append(dict(_extra_entry(n, mass_len, 0, 1, args.steps, 0.0, None, world),
                                  skipped="host transport: a host callback cannot run inside a captured graph"))
            sim.close()
            sim = None
            if guard:
                guard.disarm()
            if gasp:
                gasp.disarm()
    elif not sharded and not args.no_extras and not args.dry_run:
        # same K steps twice more (run-to-run spread), then the LDS-tile route of the north star on the same chain
        solo_leg("repeats")
        put("repeat_ms_per_step", [timed_leg(sim, args.steps, 0) / args.steps * 1e3 for _ in range(2)])
        if not args.no_clock_probe and rank == 0:
            # the clock the chip holds UNDER THE STEP KERNEL: the same K steps once more with the sampler running beside them
            solo_leg("clock sampler leg")
            try:
                nb.clock_sampler_begin(0.5, 3.0 * elapsed * 1e3 + 500.0)
                e_clk = timed_leg(sim, args.steps, 0)
                clk_ms, clk_launches = sim.last_step_ms()
                sampled = nb.clock_sampler_end()
                clk_s = clk_ms * 1e-3 / max(clk_launches, 1)
                sampled["leg"] = {"ms_per_step": e_clk / args.steps * 1e3, "kernel_ms_per_launch": clk_s * 1e3,
                                  "cycles_per_wave_interaction": clk_s * sampled["clock_ghz"] * 1e9 * 4 * device_cus(info)
                                  / (launch_interactions / 64.0) if sampled.get("clock_ghz") else None}
                with out_lock:
                    out["roofline"].update(held_clock_fields(clock, sampled, per_launch_s, launch_interactions, info, achieved_tflops))
            except Exception as e:  # pragma: no cover - diagnostic only
                put("clock_sampler_error", str(e))
        solo_leg("alt_lds")
        sim.configure(variant=0)
        e_lds = timed_leg(sim, args.steps, 2)
        lds_ms, lds_launches = sim.last_step_ms()
        lds_s = lds_ms * 1e-3 / max(lds_launches, 1)
        lds_tf = float(n) * float(mass_len) * (args.steps / max(lds_launches, 1)) * FLOP_PER_INTERACTION / lds_s / 1e12
        if rank == 0:
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
            solo_leg("extra_configs C2/C3/N2/C1")
            configs = single_gpu_configs(nb, stamp=not args.no_parity)
            put("extra_configs", configs)
            solo_leg("extra_configs S2/S4/S8/C5S8")
            try:
                configs.extend(shard_scaling_configs(nb, elapsed / args.steps * 1e3, stamp=not args.no_parity, n5=args.n5))
            except Exception as e:  # pragma: no cover - diagnostic only
                configs.append({"config": "S2/S4/S8/C5S8", "error": str(e)})
    if sim is not None:
        sim.close()
    if solo_gasp:
        solo_gasp.disarm()

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


XGMI_LINK_GBS = 153.0            # per direction and link, 7 links per GPU (task brief)
GATHER_LATENCY_ASSUMED_MS = 0.05  # fixed cost of one small in-stream all-gather: an ASSUMPTION, never measured on > 1 device here


def gather_estimate_ms(mass_chunk, ranks):
    """What one per-step all-gather of `mass_chunk` float2 per rank should cost across `ranks` GPUs: every slice rides its
    own xGMI link (direct all-gather), plus an assumed fixed latency.  An estimate with its source stated --
    the only multi-GPU term of the curve that a single-GPU box cannot measure."""
    wire = mass_chunk * 8.0 / (XGMI_LINK_GBS * 1e9) * 1e3 if ranks > 1 else 0.0
    return wire + (GATHER_LATENCY_ASSUMED_MS if ranks > 1 else 0.0)


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


# ---- N > 1: a GPU-free supervisor over fresh rank processes --------------------------------------------------------------

RUNNING = -1000   # status of a rank process that has not ended yet (exit codes and -signal numbers are > -1000)


def _die_with_parent():
    """In the child, between fork and exec: a rank never outlives the supervisor that started it, however that one ends
    (PR_SET_PDEATHSIG survives the exec; the signal handlers of supervise() cover the polite ways of being stopped)."""
    try:
        C.CDLL(None).prctl(1, 9)   # PR_SET_PDEATHSIG, SIGKILL
    except Exception:
        pass


class RankProcess:
    """One worker (this file, NB_BENCH_WORKER=1) for one rank of one attempt, started by a process that never touches the
    GPU.  Rank 0's stdout (the JSON line) is captured; every worker's stderr is forwarded as it comes and its tail kept."""

    def __init__(self, argv, env, rank, capture_stdout):
        self.rank = rank
        self.lines, self.tail = [], []
        self.proc = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, preexec_fn=_die_with_parent,
                                     stdout=subprocess.PIPE if capture_stdout else sys.stderr.fileno(), stderr=subprocess.PIPE)
        self.threads = [threading.Thread(target=self._pump_err, daemon=True)]
        if capture_stdout:
            self.threads.append(threading.Thread(target=self._pump_out, daemon=True))
        for t in self.threads:
            t.start()

    def _pump_out(self):
        for raw in self.proc.stdout:
            self.lines.append(raw.decode(errors="replace"))

    def _pump_err(self):
        for raw in self.proc.stderr:
            text = raw.decode(errors="replace")
            sys.stderr.write(text)
            sys.stderr.flush()
            self.tail.append(text)
            del self.tail[:-40]

    def status(self):
        rc = self.proc.poll()
        return RUNNING if rc is None else rc

    def end(self):
        """By exact pid: this Popen's own child, nothing matched by name."""
        if self.proc.poll() is None:
            self.proc.kill()

    def finish(self):
        self.proc.wait()
        for t in self.threads:
            t.join(5.0)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _headline_of(lines, world, dry_run):
    """(line dict or None, why not): the last JSON object rank 0 wrote, accepted when it is a complete headline -- metric and
    value present and, for a real multi-rank run, a self-check that passed."""
    for text in reversed(lines):
        text = text.strip()
        if not text.startswith("{"):
            continue
        try:
            line = json.loads(text)
        except ValueError:
            continue
        if "metric" not in line or "value" not in line:
            return None, "rank 0 wrote a line without metric / value"
        if world > 1 and not dry_run:
            check = line.get("self_check")
            if not check:
                return None, "the line carries no self_check"
            if not check.get("ranks_agree") or check.get("ok") is False:
                return line, "self_check failed: " + json.dumps(check)
        return line, None
    return None, "rank 0 wrote no JSON line"


class Supervisor:
    """`bench.py --gpus N` (N > 1) started bare, or started once per rank by torch.distributed.run: THIS process never makes
    a GPU call.  Bare: it starts N fresh rank processes itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT in their
    environment) -- the shape of the reference harness, one plain command (src/bench.c:41-74).  Under torch.distributed.run:
    every rank process supervises ONE fresh worker and the supervisors keep each other informed over their own rank link.
    Either way an attempt whose ranks do not all deliver -- RCCL that does not come up (the library's watchdog leaves with 3,
    its error convention with abort()), a self-check that fails -- is followed, with --transport auto, by a SECOND attempt in
    fresh processes over the direct exchange; the line then carries "transport_fallback".  Nothing is ever retried or
    re-executed inside a process that has touched the GPU; stragglers are ended by exact pid, and no rank outlives the
    supervisor that started it.
    Exit code: what the ranks of the LAST attempt left with -- 0 only when every one of them did; a run whose optional leg
    stalled (4) or aborted (6) after the headline still writes the complete line ("extras_aborted",
    launch.attempts[].child_rcs) and still does not report success."""

    GRACE_S = 20.0   # what the other ranks get once one has left with an error (rank 0 may be writing its line)

    def __init__(self, args, argv):
        self.args = args
        under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ
        if under_launcher:
            self.rank, self.world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
            self.local = [(self.rank, int(os.environ.get("LOCAL_RANK", self.rank)))]
            self.port = os.environ["MASTER_PORT"]
            self.mode = "torch.distributed.run: every rank process stays GPU-free and supervises one fresh worker"
        else:
            self.rank, self.world = 0, args.gpus
            self.local = [(r, r) for r in range(self.world)]
            self.port = str(_free_port())
            self.mode = f"bare: bench.py started its {self.world} rank processes itself"
        self.run_id = os.environ.get("TORCHELASTIC_RUN_ID", "none")
        self.link = None
        if under_launcher and self.world > 1:
            from nbody_amd.ranklink import RankLink
            self.link = RankLink(self.rank, self.world, name=f"nbody_sup_{self.port}_{self.run_id}")
        # what the workers get: the same command line minus what the supervisor decides
        self.passthrough, skip = [], False
        for a in argv:
            if skip:
                skip = False
            elif a == "--transport":
                skip = True
            elif not a.startswith("--transport="):
                self.passthrough.append(a)
        self.transports = ["rccl", "direct"] if args.transport == "auto" else [args.transport]
        self.ranks, self.attempts = [], []

    def everyone(self, values):
        """Status of every rank, indexed by rank (bare: they are all mine)."""
        if self.link is None:
            return list(values)
        return [int(x) for row in self.link.allgather([float(v) for v in values]) for x in row]

    def end_my_ranks(self):
        """Whatever ends this supervisor early -- a launcher's SIGTERM, an exception on the supervisors' link -- must not
        leave rank processes behind on the GPUs: end exactly the children this process started."""
        for p in self.ranks:
            p.end()

    def attempt(self, index, transport):
        """One set of fresh rank processes over `transport`: (line or None, why it does not count or None, record)."""
        env = dict(os.environ, NB_BENCH_WORKER="1", WORLD_SIZE=str(self.world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(self.port),
                   NB_BENCH_ATTEMPT=str(index), TORCHELASTIC_RUN_ID=self.run_id)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if len(self.transports) > 1 and index == 0:
            env.setdefault("NB_HIP_COMM_TIMEOUT_S", "75")   # there is a fallback: do not sit out the library's 180 s
        t0 = time.monotonic()
        self.ranks[:] = [RankProcess(self.passthrough + ["--transport", transport], dict(env, RANK=str(r), LOCAL_RANK=str(lr)), r, r == 0)
                         for r, lr in self.local]
        first_failure = None
        while True:
            seen = self.everyone([p.status() for p in self.ranks])
            if all(v != RUNNING for v in seen):
                break
            now = time.monotonic()
            if first_failure is None and any(v not in (RUNNING, 0) for v in seen):
                first_failure = now
            # a rank that left with an error takes the attempt with it: the others get a moment (rank 0 may be writing
            # its line; the library's own watchdogs may still fire), then go -- by exact pid
            if (first_failure is not None and now - first_failure > self.GRACE_S) or now - t0 > self.args.attempt_timeout_s:
                self.end_my_ranks()
            time.sleep(0.25)
        for p in self.ranks:
            p.finish()
        rcs = self.everyone([p.status() for p in self.ranks])
        record = {"transport": transport, "child_rcs": rcs, "seconds": round(time.monotonic() - t0, 2)}
        # every rank's last words, indexed by rank (under a launcher each supervisor holds one worker's)
        tails = ["".join(p.tail)[-900:] for p in self.ranks]
        if self.link is not None:
            tails = [t.decode(errors="replace") for t in self.link.allgather(tails[0].encode())]
        line, why = None, None
        if self.rank == 0:
            line, why = _headline_of(self.ranks[0].lines, self.world, self.args.dry_run)
            if line is None or why is not None:
                # whose stderr explains it: a rank that left with something other than Python's generic 1, if there is one
                bad = sorted(range(self.world), key=lambda r: (rcs[r] == 0, rcs[r] == 1))[0]
                record["why_not"] = why
                record["stderr_tail"] = f"[rank {bad}, rc {rcs[bad]}] " + tails[bad]
        good = 1 if (line is not None and why is None) else 0
        if self.link is not None:
            good = int(self.link.broadcast(good if self.rank == 0 else None, src=0))
        self.attempts.append(record)
        return line, why, bool(good)

    def run(self):
        line, why, good = None, "no attempt ran", False
        for index, transport in enumerate(self.transports):
            line, why, good = self.attempt(index, transport)
            if good:
                break
        if self.link is not None:
            self.link.barrier()
            self.link.close()
        last = [rc for rc in self.attempts[-1]["child_rcs"] if rc not in (0, RUNNING)]
        code = (min(abs(last[0]), 255) or 1) if last else 0      # the worst the final attempt's ranks left with; 0 only when all did
        if self.rank != 0:
            return code if good else (code or 1)
        launch = {"mode": self.mode, "attempts": self.attempts}
        if good:
            line["launch"] = launch
            if len(self.attempts) > 1:
                first = self.attempts[0]
                line["transport_fallback"] = {"from": first["transport"], "to": self.attempts[-1]["transport"], "rc": first["child_rcs"],
                                              "why": first.get("why_not"), "stderr_tail": first.get("stderr_tail", "")[-600:]}
            print(json.dumps(line), flush=True)
            return code
        # no complete headline from any attempt: still one line, saying so
        partial = line if isinstance(line, dict) else {}
        partial.update({"metric": partial.get("metric", "particle-pair interactions/sec at N=2^20"), "value": partial.get("value"),
                        "unit": "interactions/s", "n_gpus": self.world, "error": why, "launch": launch})
        print(json.dumps(partial), flush=True)
        return code or 1


def supervise(args, argv):
    import signal
    sup = Supervisor(args, argv)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, lambda n, f: (sup.end_my_ranks(), os._exit(128 + n)))
    try:
        return sup.run()
    finally:
        sup.end_my_ranks()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if os.environ.get("NB_BENCH_WORKER") == "1" or (args.gpus <= 1 and not launched):
        worker_main(args)     # one rank; the only place a GPU is touched
        return 0
    return supervise(args, argv)


if __name__ == "__main__":
    sys.exit(main())
