"""ctypes view of oracle/liboracle.so -- the parity CHECKER (test infrastructure only).

Particles travel as float32 arrays of shape (n, 8): pos.xy vel.xy acc.xy mass radius,
byte-identical to `Particle[n]` (reference include/nbody.h:47-50).
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_CPU_SO = os.path.join(ORACLE_DIR, "_ref", "libnbody_ref_cpu.so")
REF_WORLD_SO = os.path.join(ORACLE_DIR, "_ref", "libnbody_ref_world.so")

_lib = None


def build():
    """(Re)build the checker; cheap no-op when up to date."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True)


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "nbody_oracle.c")
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
            build()
        L = C.CDLL(ORACLE_SO)
        L.orc_partition.restype = C.c_uint32
        L.orc_partition.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_partition_ints.restype = C.c_uint32
        L.orc_partition_ints.argtypes = [C.c_void_p, C.c_uint32]
        for name in ("orc_step_avx_order", "orc_step_avx", "orc_step_seq", "orc_step_f64"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32]
        L.orc_step_lanes.restype = None
        L.orc_step_lanes.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32]
        L.orc_acc_f64.restype = None
        L.orc_acc_f64.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_acc_f64_subset.restype = None
        L.orc_acc_f64_subset.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_acc_avx_subset.restype = None
        L.orc_acc_avx_subset.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_time_avx_sample.restype = C.c_double
        L.orc_time_avx_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                                          C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _check(a):
    assert a.dtype == np.float32 and a.ndim == 2 and a.shape[1] == 8 and a.flags.c_contiguous
    return a


def partition(a):
    """world.c:32-46 on a copy; returns (partitioned, mass_len)."""
    out = _check(np.ascontiguousarray(a, dtype=np.float32)).copy()
    m = lib().orc_partition(out.ctypes.data, out.shape[0])
    return out, int(m)


def partition_ints(v):
    arr = np.asarray(v, dtype=np.int32).copy()
    m = lib().orc_partition_ints(arr.ctypes.data, arr.shape[0])
    return arr.tolist(), int(m)


def step(a, mass_len, dt, n, kind="avx_order"):
    """n steps on a copy of partitioned particles; kind in avx_order|avx|seq|f64."""
    out = _check(np.ascontiguousarray(a, dtype=np.float32)).copy()
    getattr(lib(), "orc_step_" + kind)(out.ctypes.data, out.shape[0], mass_len, dt, n)
    return out


def step_lanes(a, mass_len, dt, n, lanes):
    """n steps in the summation order of the reference build with `lanes`-wide packs (8 AVX, 4 SSE, 1 scalar)."""
    out = _check(np.ascontiguousarray(a, dtype=np.float32)).copy()
    lib().orc_step_lanes(out.ctypes.data, out.shape[0], mass_len, dt, n, lanes)
    return out


def acc_f64(a, mass_len):
    """float64 accelerations of the current state and per-component sum of |contributions|."""
    a = _check(np.ascontiguousarray(a, dtype=np.float32))
    n = a.shape[0]
    acc = np.zeros((n, 2), dtype=np.float64)
    mag = np.zeros((n, 2), dtype=np.float64)
    lib().orc_acc_f64(a.ctypes.data, n, mass_len, acc.ctypes.data, mag.ctypes.data)
    return acc, mag


def acc_f64_subset(a, mass_len, idx):
    """float64 accelerations (and sum of |contributions|) of the receivers idx only."""
    a = _check(np.ascontiguousarray(a, dtype=np.float32))
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    acc = np.zeros((idx.size, 2), dtype=np.float64)
    mag = np.zeros((idx.size, 2), dtype=np.float64)
    lib().orc_acc_f64_subset(a.ctypes.data, mass_len, idx.ctypes.data, idx.size, acc.ctypes.data, mag.ctypes.data)
    return acc, mag


def acc_avx_subset(a, mass_len, idx):
    """fp32 accelerations the reference AVX path would produce for the receivers idx (bit-exact order)."""
    a = _check(np.ascontiguousarray(a, dtype=np.float32))
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    acc = np.zeros((idx.size, 2), dtype=np.float32)
    lib().orc_acc_avx_subset(a.ctypes.data, mass_len, idx.ctypes.data, idx.size, acc.ctypes.data)
    return acc


def time_avx_sample(a, mass_len, recv_begin, recv_end, dt=0.01, threads=0):
    a = _check(np.ascontiguousarray(a, dtype=np.float32))
    used = C.c_int(0)
    chk = C.c_double(0)
    sec = lib().orc_time_avx_sample(a.ctypes.data, mass_len, recv_begin, recv_end, dt, threads,
                                    C.byref(used), C.byref(chk))
    return float(sec), int(used.value), float(chk.value)


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
