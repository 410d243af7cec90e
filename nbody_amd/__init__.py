"""nbody_amd -- Python view of the MI355X-native N-body engine (for tests, bench.py and tooling).

The product is C: `lib/libnbody_hip.so` (HIP kernels + the C-ABI of include/nbody_hip.h) and
`lib/libnbody.so` (the include/nbody.h / galaxy.h surface, C host code).  This module only binds
them with ctypes; no arithmetic happens in Python and there is no fallback: if the libraries are
missing, importing the bound functions raises, and any GPU call without a gfx950 device aborts
inside the library (reference error convention, src/lib/util.h:17-29).

Particles travel as float32 arrays of shape (n, 8): pos.xy vel.xy acc.xy mass radius --
byte-identical to `Particle[n]` (include/nbody.h).
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_DIR = os.path.join(PKG_DIR, "lib")
HIP_SO = os.environ.get("NBODY_HIP_SO") or os.path.join(LIB_DIR, "libnbody_hip.so")  # override: kernel experiments
NBODY_SO = os.path.join(LIB_DIR, "libnbody.so")



def _nb_g_from_header():
    """NB_G as include/nbody.h spells it -- the one place the value is written down."""
    import re
    with open(os.path.join(ROOT, "include", "nbody.h")) as f:
        return float(re.search(r"^#define\s+NB_G\s+([0-9.eE+-]+)f?\s*$", f.read(), re.M).group(1))


NB_G = _nb_g_from_header()
UNIQUE_ID_BYTES = 128
CLOCK_SAMPLER_MAX_MS = 20000.0   # include/nbody_hip.h NB_CLOCK_SAMPLER_MAX_MS: larger bounds are clamped by the library


class WorldData(C.Structure):
    """include/nbody_hip.h WorldData (reference src/lib/sim_gpu.h:8-12)."""
    _fields_ = [("total_len", C.c_uint32), ("mass_len", C.c_uint32), ("dt", C.c_float)]


class NbShardPlan(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("mass_chunk", "zero_chunk", "mass_begin", "mass_count",
                                          "zero_begin", "zero_count", "src_padded")]


# include/nbody_hip.h NbAllGatherFn: (ctx, buf, bytes_per_rank, rank, nranks)
ALLGATHER_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int)

# every symbol include/nbody_hip.h declares: (restype, argtypes)
HIP_API = {
    "CreateSimPipeline": (C.c_void_p, [WorldData]),
    "DestroySimPipeline": (None, [C.c_void_p]),
    "GetSimulationData": (None, [C.c_void_p, C.c_void_p]),
    "SetSimulationData": (None, [C.c_void_p, C.c_void_p]),
    "PerformSimUpdate": (None, [C.c_void_p, C.c_uint32, C.c_float]),
    "nb_hip_device_count": (C.c_int, []),
    "nb_hip_set_device": (None, [C.c_int]),
    "nb_hip_device_info": (None, [C.c_char_p, C.c_uint32]),
    "nb_hip_step_async": (None, [C.c_void_p, C.c_uint32, C.c_float]),
    "nb_hip_sync": (None, [C.c_void_p]),
    "nb_hip_last_step_ms": (C.c_double, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "nb_hip_last_finish_launches": (C.c_uint32, [C.c_void_p]),
    "nb_hip_last_step_breakdown": (C.c_uint32, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "nb_hip_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_char_p, C.c_uint32]),
    "nb_hip_comm_bringup": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "nb_hip_preflight_peers": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "nb_hip_preflight_ipc_export": (C.c_int, [C.c_void_p, C.c_uint32]),
    "nb_hip_preflight_ipc_open": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]),
    "nb_hip_preflight_ipc_release": (None, []),
    "nb_hip_error_string": (C.c_char_p, [C.c_int]),
    "nb_hip_graph_stats": (C.c_uint32, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "nb_hip_runtime_version": (C.c_int, []),
    "nb_hip_probe_clock": (C.c_int, [C.c_double] + [C.POINTER(C.c_double)] * 5),
    "nb_hip_clock_sampler_begin": (C.c_int, [C.c_double, C.c_double]),
    "nb_hip_clock_sampler_end": (C.c_int, [C.POINTER(C.c_double)] * 6 + [C.POINTER(C.c_uint32)]),
    "nb_hip_note_host_array": (None, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "nb_hip_configure": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "nb_hip_launch_shape": (None, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), C.POINTER(C.c_uint32)]),
    "nb_hip_plan_launch": (None, [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(C.c_int), C.POINTER(C.c_uint32)]),
    "nb_hip_comm_unique_id": (None, [C.c_void_p]),
    "CreateSimPipelineSharded": (C.c_void_p, [WorldData, C.c_int, C.c_int, C.c_void_p]),
    "CreateSimPipelineShardedWith": (C.c_void_p, [WorldData, C.c_int, C.c_int, ALLGATHER_FN, C.c_void_p]),
    "CreateSimPipelineShardedDirect": (C.c_void_p, [WorldData, C.c_int, C.c_int, ALLGATHER_FN, C.c_void_p]),
    "nb_hip_shard_plan": (NbShardPlan, [C.c_uint32, C.c_uint32, C.c_int, C.c_int]),
    "nb_hip_local_group_create": (C.c_int, [WorldData, C.c_int, C.POINTER(C.c_void_p)]),
    "nb_hip_local_group_step": (None, [C.POINTER(C.c_void_p), C.c_int, C.c_uint32, C.c_float]),
    "nb_hip_version": (C.c_int, []),
}

# nbody_amd/csrc/nbody_hip_tuning.h: test and tooling hooks, exported by the library but NOT part of the C-ABI
TUNE_API = {
    "nb_hip_tune": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "nb_hip_tuning_build": (C.c_int, []),
    "nb_hip_launch_unit": (C.c_int, [C.c_void_p]),
    "nb_hip_last_fused_steps": (C.c_uint32, [C.c_void_p]),
    "nb_hip_launch_lanes": (C.c_int, [C.c_void_p]),
    "nb_hip_plan_launch_unit": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int]),
    "nb_hip_plan_fused_finish": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int]),
    "nb_hip_plan_launch_lanes": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_int)]),
}
PUBLIC_KNOBS = ("variant", "graph", "timing", "overlap", "sharded_graph")   # nb_hip_configure; everything else is a tuning hook

# include/nbody.h + include/galaxy.h
NBODY_API = {
    "CreateWorld": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "DestroyWorld": (None, [C.c_void_p]),
    "GetWorldParticles": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "UpdateWorld_CPU": (None, [C.c_void_p, C.c_float, C.c_uint32]),
    "UpdateWorld_GPU": (None, [C.c_void_p, C.c_float, C.c_uint32]),
    "CreateWorldSharded": (C.c_void_p, [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]),
    "CreateWorldShardedWith": (C.c_void_p, [C.c_void_p, C.c_uint32, C.c_int, C.c_int, ALLGATHER_FN, C.c_void_p]),
    "CreateWorldShardedDirect": (C.c_void_p, [C.c_void_p, C.c_uint32, C.c_int, C.c_int, ALLGATHER_FN, C.c_void_p]),
    "GetWorldPipeline": (C.c_void_p, [C.c_void_p]),
    "MakeGalaxies": (C.c_void_p, [C.c_uint32, C.c_uint32]),
    "MakeGalaxiesSeeded": (C.c_void_p, [C.c_uint32, C.c_uint32, C.c_uint64]),
}

_hip = None
_nbody = None


def _bind(lib, api):
    for name, (res, args) in api.items():
        f = getattr(lib, name)  # AttributeError if the library does not export it
        f.restype = res
        f.argtypes = args
    return lib


def _build_if_missing(path):
    """A missing library is built once, loudly; there is no other implementation to fall back to."""
    if os.path.exists(path) or os.environ.get("NBODY_HIP_SO"):
        return
    import sys
    from . import build as _build
    print(f"[nbody_amd] {path} missing: building the product libraries (hipcc, gfx950)", file=sys.stderr, flush=True)
    _build.build_product()


def hip_lib():
    """libnbody_hip.so, loaded once.  Raises OSError when it cannot be built or loaded."""
    global _hip
    if _hip is None:
        _build_if_missing(HIP_SO)
        if not os.path.exists(HIP_SO):
            raise OSError(f"{HIP_SO} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _hip = _bind(_bind(C.CDLL(HIP_SO, mode=C.RTLD_GLOBAL), HIP_API), TUNE_API)
    return _hip


def nbody_lib():
    """libnbody.so (World API), loaded once; pulls libnbody_hip.so in first."""
    global _nbody
    if _nbody is None:
        hip_lib()
        _build_if_missing(NBODY_SO)
        if not os.path.exists(NBODY_SO):
            raise OSError(f"{NBODY_SO} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _nbody = _bind(C.CDLL(NBODY_SO, mode=C.RTLD_GLOBAL), NBODY_API)
    return _nbody


def as_particles(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 8:
        raise ValueError("particles must have shape (n, 8)")
    return a


def device_count():
    return int(hip_lib().nb_hip_device_count())


def device_info():
    buf = C.create_string_buffer(256)
    hip_lib().nb_hip_device_info(buf, 256)
    return buf.value.decode()


def probe_clock(target_ms=40.0):
    """include/nbody_hip.h nb_hip_probe_clock: the shader clock held under the step kernels' instruction mix."""
    v = [C.c_double(0.0) for _ in range(5)]
    waves = hip_lib().nb_hip_probe_clock(float(target_ms), *[C.byref(x) for x in v])
    return {"clock_ghz": v[0].value, "clock_ghz_min": v[1].value, "clock_ghz_max": v[2].value,
            "cycles_per_wave_interaction": v[3].value, "elapsed_ms": v[4].value, "waves": int(waves)}


def clock_sampler_begin(period_ms=0.5, max_ms=6000.0):
    """include/nbody_hip.h nb_hip_clock_sampler_begin: sample the shader clock while other kernels run."""
    return int(hip_lib().nb_hip_clock_sampler_begin(float(period_ms), float(max_ms)))


def clock_sampler_end():
    ghz, lo, hi, span = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    per_xcd, profile = (C.c_double * 8)(), (C.c_double * 10)()
    dropped = C.c_uint32(0)
    n = hip_lib().nb_hip_clock_sampler_end(C.byref(ghz), C.byref(lo), C.byref(hi), per_xcd, profile, C.byref(span), C.byref(dropped))
    return {"clock_ghz": ghz.value, "clock_ghz_min": lo.value, "clock_ghz_max": hi.value, "per_xcd_ghz": [float(v) for v in per_xcd],
            "profile_ghz": [float(v) for v in profile], "span_ms": span.value, "intervals": int(n), "dropped_intervals": int(dropped.value)}


def preflight_peers(max_devices=16):
    """include/nbody_hip.h nb_hip_preflight_peers: (visible devices, hipDeviceCanAccessPeer row of this process' device)."""
    row = (C.c_int * max_devices)()
    count = int(hip_lib().nb_hip_preflight_peers(row, max_devices))
    return count, [int(v) for v in row[:min(count, max_devices)]]


def preflight_ipc_export(tag):
    """(hipError_t, 64-byte IPC handle of a device word holding `tag`)."""
    buf = (C.c_ubyte * 64)()
    rc = int(hip_lib().nb_hip_preflight_ipc_export(buf, int(tag) & 0xffffffff))
    return rc, bytes(buf)


def preflight_ipc_open(handle, expect_tag):
    """(0 / hipError_t / -1, host ms) of mapping a peer's exported word, reading it and unmapping."""
    ms = C.c_double(0.0)
    buf = (C.c_ubyte * 64).from_buffer_copy(handle)
    rc = int(hip_lib().nb_hip_preflight_ipc_open(buf, int(expect_tag) & 0xffffffff, C.byref(ms)))
    return rc, float(ms.value)


def hip_error_string(code):
    return hip_lib().nb_hip_error_string(int(code)).decode(errors="replace")


def shard_plan(total_len, mass_len, rank, nranks):
    p = hip_lib().nb_hip_shard_plan(total_len, mass_len, rank, nranks)
    return {n: int(getattr(p, n)) for n, _ in NbShardPlan._fields_}


def plan_launch(n_recv, n_src, compute_units=256):
    """The classic (k, w, split, unit) plan, plus "lanes" / "lanes_w": whether an all-auto unsharded step of that size
    runs as a lane-split launch instead (lanes > 1) and with how many waves per workgroup."""
    k, w, sp, g, lw = C.c_int(), C.c_int(), C.c_int(), C.c_uint32(), C.c_int()
    hip_lib().nb_hip_plan_launch(n_recv, n_src, compute_units, C.byref(k), C.byref(w), C.byref(sp), C.byref(g))
    lanes = int(hip_lib().nb_hip_plan_launch_lanes(n_recv, n_src, C.byref(lw)))
    return {"k": k.value, "w": w.value, "split": sp.value, "workgroups": g.value,
            "unit": int(hip_lib().nb_hip_plan_launch_unit(n_recv, n_src, compute_units)), "lanes": lanes, "lanes_w": lw.value,
            "fused_finish": int(hip_lib().nb_hip_plan_fused_finish(n_recv, n_src, compute_units))}


def comm_unique_id():
    buf = (C.c_ubyte * UNIQUE_ID_BYTES)()
    hip_lib().nb_hip_comm_unique_id(buf)
    return bytes(buf)


class SimPipeline:
    """The inner seam (reference src/lib/sim_gpu.h:21-36) as an object.

    `particles` passed to set_data must already be partitioned (mass > 0 first) with
    `mass_len` massive ones, exactly what reference src/lib/world.c:32-58 hands its backend.
    """

    def __init__(self, total_len, mass_len, rank=0, nranks=1, unique_id=None, allgather=None, direct=False):
        """allgather: a Python callable (buf: writable uint8 array of shape (nranks, bytes_per_rank), rank, nranks) that
        fills every row with its owner's bytes -- the caller-supplied host transport (CreateSimPipelineShardedWith);
        direct=True: the same callable carries only IPC handles and step barriers, the data goes device to device
        (CreateSimPipelineShardedDirect)."""
        L = hip_lib()
        wd = WorldData(total_len, mass_len, 0.0)
        self._cb = None
        if allgather is not None:
            def thunk(_ctx, buf, bytes_per_rank, r, n):
                # An exception must not escape into ctypes (it would be printed and swallowed, the staging buffer would
                # keep stale peer slots, and this rank would step on them while the others block in their next
                # collective).  The process has touched the GPU: report and leave with a fresh exit, no retry.
                try:
                    rows = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(n, int(bytes_per_rank)))
                    allgather(rows, r, n)
                except BaseException:
                    import sys
                    import traceback
                    traceback.print_exc()
                    print(f"[nbody_amd] rank {r} of {n}: the caller-supplied all-gather raised; exiting (5)", file=sys.stderr,
                          flush=True)
                    os._exit(5)
            self._cb = ALLGATHER_FN(thunk)   # must outlive the pipeline
            create = L.CreateSimPipelineShardedDirect if direct else L.CreateSimPipelineShardedWith
            self._h = create(wd, rank, nranks, self._cb, None)
        elif nranks > 1 or unique_id is not None:
            idbuf = (C.c_ubyte * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
            self._h = L.CreateSimPipelineSharded(wd, rank, nranks, idbuf)
        else:
            self._h = L.CreateSimPipeline(wd)
        self.total_len, self.mass_len = total_len, mass_len
        self.rank, self.nranks = rank, nranks
        self.configure(timing=1)   # tooling wants last_step_ms(); the C default is off (it costs a frame loop 3-7 us)

    def close(self):
        if self._h:
            hip_lib().DestroySimPipeline(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_data(self, particles):
        a = as_particles(particles)
        assert a.shape[0] == self.total_len
        hip_lib().SetSimulationData(self._h, a.ctypes.data)

    def get_data(self):
        out = np.empty((self.total_len, 8), dtype=np.float32)
        hip_lib().GetSimulationData(self._h, out.ctypes.data)
        return out

    def update(self, n, dt):
        """Blocking n steps (PerformSimUpdate)."""
        hip_lib().PerformSimUpdate(self._h, n, dt)

    def step_async(self, n, dt):
        hip_lib().nb_hip_step_async(self._h, n, dt)

    def sync(self):
        hip_lib().nb_hip_sync(self._h)

    def last_step_ms(self):
        launches = C.c_uint32(0)
        ms = hip_lib().nb_hip_last_step_ms(self._h, C.byref(launches))
        return float(ms), int(launches.value)

    def finish_launches(self):
        return int(hip_lib().nb_hip_last_finish_launches(self._h))

    def step_breakdown(self):
        """(steps covered, kernel ms, all-gather ms) of the last update of a sharded pipeline."""
        k, c = C.c_double(0.0), C.c_double(0.0)
        steps = hip_lib().nb_hip_last_step_breakdown(self._h, C.byref(k), C.byref(c))
        return int(steps), float(k.value), float(c.value)

    def comm_info(self):
        """What the RCCL communicator itself reports (ncclCommCount etc.); owns_comm False when there is none."""
        n, r, d, v, ms = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
        path = C.create_string_buffer(256)
        own = hip_lib().nb_hip_comm_info(self._h, C.byref(n), C.byref(r), C.byref(d), C.byref(v), C.byref(ms), path, 256)
        return {"owns_comm": bool(own), "nranks": n.value, "rank": r.value, "device": d.value, "rccl_version": v.value,
                "first_gather_ms": ms.value, "rccl_lib": path.value.decode()}

    def comm_bringup(self):
        """nb_hip_comm_bringup: ncclCommInitRank host ms, first all-gather device ms, one warm 8-byte all-gather in device us."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        own = hip_lib().nb_hip_comm_bringup(self._h, C.byref(a), C.byref(b), C.byref(c))
        return {"owns_comm": bool(own), "comm_init_ms": a.value, "first_gather_ms": b.value, "small_gather_us": c.value}

    def graph_stats(self):
        """cached hipGraph chains and how often a new step size was written to device memory."""
        uploads = C.c_uint32(0)
        cached = hip_lib().nb_hip_graph_stats(self._h, C.byref(uploads))
        return {"cached": int(cached), "dt_uploads": int(uploads.value)}

    def fused_steps(self):
        """steps of the last update that ran inside one-workgroup chain launches (knob fused_chain)."""
        return int(hip_lib().nb_hip_last_fused_steps(self._h))

    def configure(self, **knobs):
        """The five knobs of include/nbody_hip.h go through nb_hip_configure; anything else is a launch-shape / experiment
        hook of nbody_hip_tuning.h (nb_hip_tune; aborts on an unknown name like the public call does)."""
        for k, v in knobs.items():
            fn = hip_lib().nb_hip_configure if k in PUBLIC_KNOBS else hip_lib().nb_hip_tune
            fn(self._h, k.encode(), int(v))

    def launch_shape(self):
        k, w, v, sp, g = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_uint32()
        hip_lib().nb_hip_launch_shape(self._h, C.byref(k), C.byref(w), C.byref(v), C.byref(sp), C.byref(g))
        return {"k": k.value, "w": w.value, "variant": "smem" if v.value else "lds", "split": sp.value,
                "workgroups": g.value, "unit": int(hip_lib().nb_hip_launch_unit(self._h)),
                "lanes": int(hip_lib().nb_hip_launch_lanes(self._h))}


class LocalShardGroup:
    """nranks shards of one world inside this process (include/nbody_hip.h "Local transport")."""

    def __init__(self, total_len, mass_len, nranks, **knobs):
        L = hip_lib()
        self.nranks = nranks
        self._arr = (C.c_void_p * nranks)()
        L.nb_hip_local_group_create(WorldData(total_len, mass_len, 0.0), nranks, self._arr)
        self.members = []
        for r in range(nranks):
            m = SimPipeline.__new__(SimPipeline)
            m._h = self._arr[r]
            m.total_len, m.mass_len, m.rank, m.nranks = total_len, mass_len, r, nranks
            m.configure(timing=1, **knobs)
            self.members.append(m)

    def set_data(self, particles):
        for m in self.members:
            m.set_data(particles)

    def step(self, n, dt):
        hip_lib().nb_hip_local_group_step(self._arr, self.nranks, n, dt)

    def get_data(self, rank=0):
        return self.members[rank].get_data()

    def close(self):
        for m in self.members:
            m.close()
        self.members = []


class World:
    """include/nbody.h World, bound 1:1 (CreateWorld / UpdateWorld_CPU / UpdateWorld_GPU / ...)."""

    def __init__(self, particles, rank=None, nranks=1, unique_id=None):
        """rank / nranks / unique_id: CreateWorldSharded (extension), one World per process and GPU."""
        a = as_particles(particles)
        self.size = a.shape[0]
        if rank is None:
            self._h = nbody_lib().CreateWorld(a.ctypes.data, self.size)
        else:
            idbuf = (C.c_ubyte * UNIQUE_ID_BYTES).from_buffer_copy(unique_id) if unique_id is not None else None
            self._h = nbody_lib().CreateWorldSharded(a.ctypes.data, self.size, rank, nranks, idbuf)

    def close(self):
        if self._h:
            nbody_lib().DestroyWorld(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def particles(self):
        n = C.c_uint32(0)
        p = nbody_lib().GetWorldParticles(self._h, C.byref(n))
        if n.value == 0:
            return np.empty((0, 8), dtype=np.float32)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n.value, 8)).copy()

    def pipeline(self):
        """GetWorldPipeline: the SimPipeline handle behind this World (for nb_hip_configure / the tuning hooks)."""
        return nbody_lib().GetWorldPipeline(self._h)

    def tune(self, **knobs):
        """Set knobs on the World's pipeline (nb_hip_configure for the public five, nb_hip_tune for the hooks); returns the
        previous value of each, so that a caller can see that a knob took."""
        old = {}
        for k, v in knobs.items():
            fn = hip_lib().nb_hip_configure if k in PUBLIC_KNOBS else hip_lib().nb_hip_tune
            old[k] = int(fn(self.pipeline(), k.encode(), int(v)))
        return old

    def update_cpu(self, dt, n):
        nbody_lib().UpdateWorld_CPU(self._h, dt, n)

    def update_gpu(self, dt, n):
        nbody_lib().UpdateWorld_GPU(self._h, dt, n)


_cpu_best = None


def cpu_best_lib():
    """libnbody_cpu_best.so (csrc/cpu_best.c): the informational CPU variants of sim_cpu.c; never behind UpdateWorld_CPU."""
    global _cpu_best
    if _cpu_best is None:
        path = os.path.join(LIB_DIR, "libnbody_cpu_best.so")
        _build_if_missing(path)
        lib = C.CDLL(path)
        lib.nb_cpu_variant_count.restype = C.c_int
        lib.nb_cpu_variant_name.restype = C.c_char_p
        lib.nb_cpu_variant_name.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_char_p)]
        lib.nb_cpu_variant_update.restype = C.c_int
        lib.nb_cpu_variant_update.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32]
        _cpu_best = lib
    return _cpu_best


def cpu_variants():
    """[(isa, runs on this CPU, description)] of the informational CPU steppers."""
    lib, out = cpu_best_lib(), []
    for i in range(lib.nb_cpu_variant_count()):
        ok, what = C.c_int(0), C.c_char_p()
        name = lib.nb_cpu_variant_name(i, C.byref(ok), C.byref(what))
        out.append((name.decode(), bool(ok.value), what.value.decode()))
    return out


def cpu_variant_update(isa, particles, mass_len, dt, n=1):
    """n Jacobi steps of a PARTITIONED array (sources first) with one informational CPU variant; returns the new array."""
    a = as_particles(particles).copy()
    if cpu_best_lib().nb_cpu_variant_update(isa.encode(), a.ctypes.data, a.shape[0], mass_len, dt, n) != 0:
        raise RuntimeError(f"CPU variant {isa!r} is unknown or not supported by this CPU")
    return a


def make_galaxies(particle_count, galaxy_count, seed=None, own_rng=False):
    """include/galaxy.h MakeGalaxies; `seed` calls libc srand first (bench.c:42 uses 11037).
    own_rng=True: MakeGalaxiesSeeded, the libc-independent stream (seed required)."""
    L = nbody_lib()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    if own_rng:
        p = L.MakeGalaxiesSeeded(particle_count, galaxy_count, int(seed))
    else:
        if seed is not None:
            libc.srand(C.c_uint(seed))
        p = L.MakeGalaxies(particle_count, galaxy_count)
    a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(particle_count, 8)).copy()
    libc.free(p)
    return a
