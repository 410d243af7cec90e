#!/usr/bin/env python3
"""Generate tests/golden/ fixtures from the reference's own compiled CPU path.

Runs ONLY where /root/reference exists (the build container).  It loads
oracle/_ref/libnbody_ref_cpu.so -- the reference's unmodified src/lib/sim_cpu.c
and src/lib/galaxy.c compiled by oracle/Makefile with the flags of its AVX
build -- and drives it exactly as reference src/lib/world.c:99-110 does
(PackParticles once per step, then PackedUpdate per receiver).  The partition
that precedes it (world.c:32-46) cannot be compiled stand-alone (it sits inside
CreateWorld, which needs the Vulkan side), so the fixture generator applies the
oracle's restatement of it and the result is cross-checked against the sha256
digests SURVEY.md section 8c recorded from the reference's full CreateWorld +
UpdateWorld_CPU build.

Fixtures are data only: raw little-endian Particle[] dumps (8 floats each:
pos.xy vel.xy acc.xy mass radius) and a JSON manifest with digests.
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libnbody_ref_cpu.so")
ORC_SO = os.path.join(ROOT, "oracle", "liboracle.so")

# digests of GetWorldParticles() from the reference build, SURVEY.md section 8c
SURVEY_DIGESTS = {
    "4096_partitioned": "a234f65ee03619c6a30efd4ee33856d3d448a6a108f1449d4377d9e09997b59b",
    "4096_s1_dt0.01": "33528cf11845b5c34f4350e022f258a4ac4e1ff91dce2680e35c6b58a2b1b601",
    "4096_s10_dt0.01": "cf46e5610d626546d0ae1f131219b4611a33faa57e65589286d0acd6313a806d",
    "4096_s100_dt1": "bd15994fb15d1aa43c93f609f65f2ae371817033f09f9cd712cb2534d32bfb2f",
}

# reference test/test_particle_sort.c:27-111 -- (input, expected output, expected count)
PARTITION_CASES = [
    ([1, 2, 3, 4, 5], [1, 2, 3, 4, 5], 5),
    ([0, 0, 0, 0, 0], [0, 0, 0, 0, 0], 0),
    ([1, 2, 3, 0, 0], [1, 2, 3, 0, 0], 3),
    ([0, 0, 1, 2, 3], [3, 2, 1, 0, 0], 3),
    ([0, 0, 0, 1, 2, 3], [3, 2, 1, 0, 0, 0], 3),
    ([0, 1, 2, 0, 3], [3, 1, 2, 0, 0], 3),
    ([0, 1, 2, 0, 3, 0], [3, 1, 2, 0, 0, 0], 3),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    if not os.path.exists(REF_SO):
        sys.exit("oracle/_ref/libnbody_ref_cpu.so missing: run `make -C oracle` where /root/reference exists")
    ref = C.CDLL(REF_SO)
    orc = C.CDLL(ORC_SO)
    libc = C.CDLL(None)

    ref.MakeGalaxies.restype = C.c_void_p
    ref.MakeGalaxies.argtypes = [C.c_uint32, C.c_uint32]
    ref.AllocPackArray.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32]
    ref.PackParticles.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
    ref.PackedUpdate.argtypes = [C.c_void_p, C.c_float, C.c_uint32, C.c_void_p]
    ref.FreePackArray.argtypes = [C.c_void_p]
    orc.orc_partition.restype = C.c_uint32
    orc.orc_partition.argtypes = [C.c_void_p, C.c_uint32]
    libc.free.argtypes = [C.c_void_p]

    def make_ic(n, galaxies, seed):
        libc.srand(seed)
        p = ref.MakeGalaxies(n, galaxies)
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n, 8)).copy()
        libc.free(p)
        return a

    def ref_steps(arr, m, dt, nsteps):
        """world.c:99-110 with the reference's own PackParticles/PackedUpdate (of whichever build `ref` is)."""
        arr = arr.copy()
        n = arr.shape[0]
        pack = C.c_void_p()
        plen = C.c_uint32()
        ref.AllocPackArray(C.byref(pack), C.byref(plen), m)
        base = arr.ctypes.data
        for _ in range(nsteps):
            ref.PackParticles(m, base, pack)
            for i in range(n):
                ref.PackedUpdate(base + 32 * i, dt, plen.value, pack)
        ref.FreePackArray(pack)
        return arr

    manifest = {"partition_cases": PARTITION_CASES, "sets": {}, "survey_digests": SURVEY_DIGESTS}

    # bench.c:42,53 universe: srand(11037); MakeGalaxies(N, 2).  The reference's bench makes its
    # sizes in sequence from ONE srand; here each size gets a fresh srand so files are independent.
    for n, steps in ((4096, [(1, 0.01), (10, 0.01), (100, 1.0)]), (1024, [(1, 0.01), (10, 0.01)]), (333, [(3, 0.05)])):
        ic = make_ic(n, 2, 11037)
        ic.tofile(os.path.join(HERE, f"ic_{n}.bin"))
        part = ic.copy()
        m = orc.orc_partition(part.ctypes.data, n)
        entry = {"n": n, "mass_len": int(m), "ic_sha256": sha(ic), "partitioned_sha256": sha(part), "steps": {}}
        for (k, dt) in steps:
            out = ref_steps(part, m, dt, k)
            tag = f"s{k}_dt{dt:g}"
            digest = sha(out)
            entry["steps"][tag] = {"n_steps": k, "dt": dt, "sha256": digest}
            if not (n == 4096 and k == 100):  # digest only for the chaotic 100-step case
                out.tofile(os.path.join(HERE, f"ref_avx_{n}_{tag}.bin"))
                entry["steps"][tag]["file"] = f"ref_avx_{n}_{tag}.bin"
            print(f"N={n} {tag}: {digest}")
        manifest["sets"][str(n)] = entry

    # the reference's other SIMD_SET builds (SSE: 4-wide packs, scalar: 1-wide), digests only
    manifest["simd_variants"] = {}
    for name, lanes in (("sse", 4), ("scalar", 1)):
        so = os.path.join(ROOT, "oracle", "_ref", f"libnbody_ref_cpu_{name}.so")
        if not os.path.exists(so):
            continue
        vref = C.CDLL(so)
        vref.AllocPackArray.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32]
        vref.PackParticles.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
        vref.PackedUpdate.argtypes = [C.c_void_p, C.c_float, C.c_uint32, C.c_void_p]
        vref.FreePackArray.argtypes = [C.c_void_p]
        saved, ref = ref, vref
        try:
            for n in (1024, 333):
                ic = np.fromfile(os.path.join(HERE, f"ic_{n}.bin"), dtype=np.float32).reshape(-1, 8)
                part = ic.copy()
                m = orc.orc_partition(part.ctypes.data, n)
                for (k, dt) in ((1, 0.01), (10, 0.01)):
                    digest = sha(ref_steps(part, m, dt, k))
                    manifest["simd_variants"][f"{name}_{n}_s{k}_dt{dt:g}"] = {"lanes": lanes, "n": n, "n_steps": k, "dt": dt, "sha256": digest}
                    print(f"{name} N={n} s{k}: {digest}")
        finally:
            ref = saved

    e = manifest["sets"]["4096"]
    checks = {
        "4096_partitioned": e["partitioned_sha256"],
        "4096_s1_dt0.01": e["steps"]["s1_dt0.01"]["sha256"],
        "4096_s10_dt0.01": e["steps"]["s10_dt0.01"]["sha256"],
        "4096_s100_dt1": e["steps"]["s100_dt1"]["sha256"],
    }
    for k, v in checks.items():
        ok = SURVEY_DIGESTS[k] == v
        print(("OK   " if ok else "FAIL ") + k)
        if not ok:
            sys.exit(f"digest mismatch with SURVEY.md 8c for {k}: {v}")

    # digests of this repo's own MakeGalaxiesSeeded stream are not reference outputs: carried over untouched
    try:
        with open(os.path.join(HERE, "manifest.json")) as f:
            manifest["own_rng_digests"] = json.load(f)["own_rng_digests"]
    except (OSError, KeyError):
        pass

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote", os.path.join(HERE, "manifest.json"))


if __name__ == "__main__":
    main()
