"""The include/nbody.h + galaxy.h surface on the CPU: partition, coherence rules, UpdateWorld_CPU, MakeGalaxies."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import nbody_amd as nb
import oracle_binding as ob


def tagged(masses):
    a = np.zeros((len(masses), 8), dtype=np.float32)
    a[:, 6] = masses
    a[:, 7] = 1.0
    a[:, 0] = np.arange(len(masses)) * 10.0  # x doubles as an id
    return a


def test_createworld_partition_known_answers(manifest):
    # reference test/test_particle_sort.c through the real CreateWorld (mass = the int key)
    for inp, want, count in manifest["partition_cases"]:
        w = nb.World(tagged(inp))
        got = w.particles()
        w.close()
        assert got[:, 6].astype(int).tolist() == want
        assert int((got[:, 6] > 0).sum()) == count


def test_createworld_copies_input_and_returns_partitioned_order(golden, manifest):
    ic = golden("ic_4096.bin")
    keep = ic.copy()
    w = nb.World(ic)
    ic[:] = 0  # caller's array is its own (reference world.c:30)
    got = w.particles()
    w.close()
    assert ob.sha256(got) == manifest["survey_digests"]["4096_partitioned"]
    assert np.array_equal(got, ob.partition(keep)[0])


@pytest.mark.parametrize("n", [4096, 1024, 333])
def test_update_cpu_bit_exact_with_reference(golden, manifest, n):
    e = manifest["sets"][str(n)]
    for tag, s in e["steps"].items():
        w = nb.World(golden(f"ic_{n}.bin"))
        w.update_cpu(s["dt"], s["n_steps"])
        got = w.particles()
        w.close()
        assert ob.sha256(got) == s["sha256"], f"UpdateWorld_CPU differs from the reference at N={n} {tag}"


def test_update_cpu_split_calls_equal_one_call(golden):
    w1, w2 = nb.World(golden("ic_1024.bin")), nb.World(golden("ic_1024.bin"))
    w1.update_cpu(0.01, 6)
    for _ in range(3):
        w2.update_cpu(0.01, 2)
        w2.particles()
    assert np.array_equal(w1.particles(), w2.particles())
    w1.update_cpu(0.01, 0)  # n == 0: allowed, changes nothing (reference world.c:100-109)
    assert np.array_equal(w1.particles(), w2.particles())
    w1.close(); w2.close()


def test_empty_and_tiny_worlds():
    w = nb.World(np.zeros((0, 8), dtype=np.float32))
    assert w.particles().shape == (0, 8)
    w.update_cpu(0.1, 3)
    w.close()
    one = np.array([[1, 2, 3, 4, 0, 0, 5, 1]], dtype=np.float32)
    w = nb.World(one)
    w.update_cpu(0.5, 1)
    got = w.particles()[0]
    w.close()
    # a lone massive body feels only itself at distance 0: zero force, straight line
    assert got[4] == 0 and got[5] == 0 and got[0] == 1 + 0.5 * 3 and got[1] == 2 + 0.5 * 4


def test_destroy_null_is_accepted():
    nb.nbody_lib().DestroyWorld(None)


@pytest.mark.parametrize("n", [4096, 1024, 333])
def test_make_galaxies_matches_reference_fixture(golden, manifest, n):
    got = nb.make_galaxies(n, 2, seed=11037)   # bench.c:42,53 universe
    assert ob.sha256(got) == manifest["sets"][str(n)]["ic_sha256"]
    assert got.tobytes() == golden(f"ic_{n}.bin").tobytes()


def test_make_galaxies_survey_spot_values():
    # p[0] (core) of srand(11037) MakeGalaxies(4096, 2), massless count
    a = nb.make_galaxies(4096, 2, seed=11037)
    assert a[0, 0] == 0 and a[0, 1] == 0
    assert a[0, 2] == np.float32(-232.087616) and a[0, 3] == np.float32(50.9429283)
    assert a[0, 6] == np.float32(1.06293514e10) and a[0, 7] == np.float32(438.967438)
    assert int((a[:, 6] <= 0).sum()) == 2107


@pytest.mark.skipif(not os.path.exists(ob.REF_CPU_SO), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("n,g,seed", [(6000, 3, 1), (100, 1, 7), (50000, 5, 42), (1000, 10, 3), (250, 2, 11037)])
def test_make_galaxies_bit_exact_with_compiled_reference(n, g, seed):
    ref = C.CDLL(ob.REF_CPU_SO)
    ref.MakeGalaxies.restype = C.c_void_p
    ref.MakeGalaxies.argtypes = [C.c_uint32, C.c_uint32]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.srand(seed)
    p = ref.MakeGalaxies(n, g)
    want = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n, 8)).copy()
    libc.free(p)
    assert nb.make_galaxies(n, g, seed=seed).tobytes() == want.tobytes()


@pytest.mark.parametrize("n", [4096, 1024, 333])
def test_make_galaxies_seeded_is_pinned_and_leaves_rand_alone(manifest, n):
    # the libc-independent stream (galaxy.h extension): same digests on every box, rand() state untouched
    libc = C.CDLL(None)
    libc.srand(5)
    want_next = [libc.rand() for _ in range(3)]
    libc.srand(5)
    got = nb.make_galaxies(n, 2, seed=11037, own_rng=True)
    assert [libc.rand() for _ in range(3)] == want_next
    assert ob.sha256(got) == manifest["own_rng_digests"][str(n)]
    assert nb.make_galaxies(n, 2, seed=11038, own_rng=True).tobytes() != got.tobytes()
    # and the libc mode still matches the reference afterwards
    assert ob.sha256(nb.make_galaxies(n, 2, seed=11037)) == manifest["sets"][str(n)]["ic_sha256"]


def test_make_galaxies_seeded_has_the_galaxy_h_distributions():
    a = nb.make_galaxies(200000, 2, seed=3, own_rng=True)
    mass, radius = a[:, 6], a[:, 7]
    cores = mass > 1e8
    assert cores.sum() == 2 and (radius[cores] >= 200).all() and (radius[cores] < 600).all()   # galaxy.h:13-14
    massless = mass <= 0
    assert 0.48 < massless.mean() < 0.52 and (radius[massless] == 0.5).all()                    # galaxy.c:204-206
    arms = ~cores & ~massless
    assert (radius[arms] >= 1.5).all() and (radius[arms] < 9.5).all()                            # galaxy.h:16-17
    assert np.isfinite(a).all()


def test_make_galaxies_too_few_particles_aborts():
    code = "import nbody_amd as nb; nb.make_galaxies(150, 2, seed=1); print('SURVIVED')"
    r = subprocess.run(["python", "-c", code], cwd=nb.ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and "SURVIVED" not in r.stdout and "Need at least 200 particles" in r.stderr


def test_nbody_bench_cpu_table_runs():
    exe = os.path.join(nb.LIB_DIR, "nbody-bench")
    r = subprocess.run([exe, "--cpu", "--n", "250", "--n", "500", "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, check=True)
    lines = [l.split() for l in r.stdout.strip().splitlines()]
    assert lines[0][:2] == ["N", "CPU"] and [l[0] for l in lines[1:]] == ["250", "500"]
    r = subprocess.run([exe, "--cpu", "--own-rng", "--n", "250", "--steps", "2", "--warmup", "0"],
                       capture_output=True, text=True, check=True)
    assert r.stdout.strip().splitlines()[1].split()[0] == "250"


@pytest.mark.skipif(not os.path.exists(ob.REF_WORLD_SO), reason="oracle/_ref not built")
def test_reference_world_c_runs_on_our_shim_cpu_side(golden, manifest):
    """Drop-in: the reference's OWN world.c, its sim_gpu.h symbols resolved by libnbody_hip.so."""
    nb.hip_lib()  # RTLD_GLOBAL: provides CreateSimPipeline & co.
    ref = C.CDLL(ob.REF_WORLD_SO)
    ref.CreateWorld.restype = C.c_void_p
    ref.CreateWorld.argtypes = [C.c_void_p, C.c_uint32]
    ref.GetWorldParticles.restype = C.c_void_p
    ref.GetWorldParticles.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    ref.UpdateWorld_CPU.argtypes = [C.c_void_p, C.c_float, C.c_uint32]
    ref.DestroyWorld.argtypes = [C.c_void_p]
    ic = golden("ic_4096.bin")
    w = ref.CreateWorld(ic.ctypes.data, 4096)
    ref.UpdateWorld_CPU(w, 0.01, 1)
    n = C.c_uint32()
    p = ref.GetWorldParticles(w, C.byref(n))
    got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n.value, 8)).copy()
    ref.DestroyWorld(w)
    assert ob.sha256(got) == manifest["survey_digests"]["4096_s1_dt0.01"]


@pytest.mark.parametrize("name,so", [("sse", "libnbody_sse.so"), ("scalar", "libnbody_scalar.so")])
def test_other_simd_set_builds_match_their_reference_builds(manifest, golden, name, so):
    """SIMD_SET=SSE / none builds of the World library against digests of the reference's same builds."""
    nb.hip_lib()
    lib = C.CDLL(os.path.join(nb.LIB_DIR, so))
    lib.CreateWorld.restype = C.c_void_p
    lib.CreateWorld.argtypes = [C.c_void_p, C.c_uint32]
    lib.GetWorldParticles.restype = C.c_void_p
    lib.GetWorldParticles.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    lib.UpdateWorld_CPU.argtypes = [C.c_void_p, C.c_float, C.c_uint32]
    lib.DestroyWorld.argtypes = [C.c_void_p]
    seen = 0
    for tag, e in manifest["simd_variants"].items():
        if not tag.startswith(name + "_"):
            continue
        ic = golden(f"ic_{e['n']}.bin")
        w = lib.CreateWorld(ic.ctypes.data, e["n"])
        lib.UpdateWorld_CPU(w, e["dt"], e["n_steps"])
        n = C.c_uint32()
        p = lib.GetWorldParticles(w, C.byref(n))
        got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n.value, 8)).copy()
        lib.DestroyWorld(w)
        assert ob.sha256(got) == e["sha256"], tag
        seen += 1
    assert seen == 4


@pytest.mark.parametrize("n", [333, 4096])
def test_float64_truth_build_of_the_cpu_stepper(golden, n):
    """SIMD_SET=f64 (libnbody_f64.so): every term and the sum in float64 from the fp32 inputs, one
    rounding to the fp32 acc, then the reference's fp32 integrator.  The acc equals the oracle's float64 sum rounded to
    fp32 bit for bit, and sits within the stated tolerance of the AVX build's -- the tie-breaker the tolerance names."""
    nb.hip_lib()
    lib = C.CDLL(os.path.join(nb.LIB_DIR, "libnbody_f64.so"))
    lib.CreateWorld.restype = C.c_void_p
    lib.CreateWorld.argtypes = [C.c_void_p, C.c_uint32]
    lib.GetWorldParticles.restype = C.c_void_p
    lib.GetWorldParticles.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    lib.UpdateWorld_CPU.argtypes = [C.c_void_p, C.c_float, C.c_uint32]
    lib.DestroyWorld.argtypes = [C.c_void_p]
    ic = golden(f"ic_{n}.bin")
    part, m = ob.partition(ic)
    w = lib.CreateWorld(ic.ctypes.data, n)
    lib.UpdateWorld_CPU(w, 0.01, 1)
    cnt = C.c_uint32()
    p = lib.GetWorldParticles(w, C.byref(cnt))
    got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(cnt.value, 8)).copy()
    lib.DestroyWorld(w)
    acc64, mag = ob.acc_f64(part, m)
    assert np.array_equal(got[:, 4:6], acc64.astype(np.float32))
    v = part[:, 2:4] + got[:, 4:6] * np.float32(0.01)
    assert np.array_equal(got[:, 2:4], v) and np.array_equal(got[:, 0:2], part[:, 0:2] + v * np.float32(0.01))
    avx = ob.step(part, m, 0.01, 1)
    assert np.all(np.abs(avx[:, 4:6].astype(np.float64) - acc64) <= 1e-4 * np.abs(acc64) + 1e-6 * mag)


@pytest.mark.parametrize("isa", ["avx2fma", "avx2fma_rsqrt", "avx512", "avx512_rsqrt"])
@pytest.mark.parametrize("n", [333, 4096])
def test_informational_cpu_variants_stay_within_the_fp32_tolerance(golden, isa, n):
    """libnbody_cpu_best.so (csrc/cpu_best.c; SURVEY.md 8d "best CPU" row; the reference's SIMD matrix is
    src/lib/CMakeLists.txt:24-33): sim_cpu.c built with FMA contraction / 16 lanes / rsqrt estimate + Newton.  None is
    bit-exact with a reference build -- and must not claim to be -- but each sits within the stated one-step tolerance
    of the float64 sum (1e-4 |acc| + 1e-6 sum |contribution|), integrates its own acc like the reference does up to the
    one rounding an FMA saves (contraction is allowed in these builds, as under -march=native), passes mass / radius
    through, and a variant this CPU cannot run is refused, not emulated."""
    table = {name: ok for name, ok, _ in nb.cpu_variants()}
    assert set(table) == {"avx2fma", "avx2fma_rsqrt", "avx512", "avx512_rsqrt"}
    part, m = ob.partition(golden(f"ic_{n}.bin"))
    if not table[isa]:
        with pytest.raises(RuntimeError):
            nb.cpu_variant_update(isa, part, m, 0.01, 1)
        pytest.skip(f"this CPU does not run {isa}")
    got = nb.cpu_variant_update(isa, part, m, 0.01, 1)
    acc64, mag = ob.acc_f64(part, m)
    assert np.all(np.abs(got[:, 4:6].astype(np.float64) - acc64) <= 1e-4 * np.abs(acc64) + 1e-6 * mag)
    dt64, eps = float(np.float32(0.01)), 2.0 ** -23     # one fp32 rounding of each operand's magnitude covers FMA or mul + add
    v0, a = part[:, 2:4].astype(np.float64), got[:, 4:6].astype(np.float64)
    assert np.all(np.abs(got[:, 2:4] - (v0 + a * dt64)) <= eps * (np.abs(v0) + np.abs(a * dt64)))
    p0, v1 = part[:, 0:2].astype(np.float64), got[:, 2:4].astype(np.float64)
    assert np.all(np.abs(got[:, 0:2] - (p0 + v1 * dt64)) <= eps * (np.abs(p0) + np.abs(v1 * dt64)))
    assert np.array_equal(got[:, 6:8], part[:, 6:8])
    assert not np.array_equal(got[:, 4:6], ob.step(part, m, 0.01, 1)[:, 4:6])      # informational: NOT the reference's bits
    # ten steps: relative to what the steps moved, like the GPU suite's multi-step bound
    want = ob.step(part, m, 0.01, 10)
    got10 = nb.cpu_variant_update(isa, part, m, 0.01, 10)
    p0 = part[:, 0:2].astype(np.float64)
    dg, dw = got10[:, 0:2] - p0, want[:, 0:2] - p0
    assert np.linalg.norm(dg - dw) / np.linalg.norm(dw) <= 1e-4


def test_unknown_cpu_variant_is_refused(golden):
    part, m = ob.partition(golden("ic_333.bin"))
    with pytest.raises(RuntimeError):
        nb.cpu_variant_update("avx1024", part, m, 0.01, 1)
