"""benchlib -- pure helpers of bench.py: what the host offers, what a committed profile says, and the arithmetic behind the
roofline fields.  No GPU call, no oracle, nothing timed: kept out of bench.py so that the script a driver times holds the
measurement and little else.  (The CPU baseline and the parity stamps stay in bench.py: they are the only code allowed to
use the oracle, and nothing under nbody_amd/ may.)
"""
import ctypes as C
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FLOP_PER_INTERACTION = 14        # reference op count, sim_cpu.c:169-188
PEAK_FP32_VECTOR_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (vector)"
KERNEL_SOURCES = ("nbody_amd/csrc/kernels.hip", "nbody_amd/csrc/kernels.h", "nbody_amd/csrc/interaction_asm.h")


# ---- the host ------------------------------------------------------------------------------------------------------------

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


def libgomp():
    for name in ("libgomp.so.1", "libgomp.so"):
        try:
            return C.CDLL(name)
        except OSError:
            pass
    return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None



# ---- roofline.traffic: tied to the committed PMC profile -------------------------------------------------------------------

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



# ---- the one multi-GPU term a single-GPU box cannot measure ------------------------------------------------------------------

XGMI_LINK_GBS = 153.0            # per direction and link, 7 links per GPU (task brief)
GATHER_LATENCY_ASSUMED_MS = 0.05  # fixed cost of one small in-stream all-gather: an ASSUMPTION, never measured on > 1 device here


def gather_estimate_ms(mass_chunk, ranks):
    """What one per-step all-gather of `mass_chunk` float2 per rank should cost across `ranks` GPUs: every slice rides its
    own xGMI link (direct all-gather), plus an assumed fixed latency.  An estimate with its source stated --
    the only multi-GPU term of the curve that a single-GPU box cannot measure."""
    wire = mass_chunk * 8.0 / (XGMI_LINK_GBS * 1e9) * 1e3 if ranks > 1 else 0.0
    return wire + (GATHER_LATENCY_ASSUMED_MS if ranks > 1 else 0.0)
