"""The oracle itself, pinned before anything trusts it (CPU only).

Pins: (1) the reference's own partition known-answers (test/test_particle_sort.c:27-111),
(2) fixtures produced by the reference's compiled sim_cpu.c (tests/golden/make_golden.py),
(3) the sha256 digests SURVEY.md 8c took from the reference's full UpdateWorld_CPU build.
"""
import numpy as np
import pytest

import oracle_binding as ob


def test_partition_known_answers(manifest):
    # reference test/test_particle_sort.c: all seven cases
    assert len(manifest["partition_cases"]) == 7
    for inp, want, count in manifest["partition_cases"]:
        got, m = ob.partition_ints(inp)
        assert got == want and m == count


def test_partition_particles_matches_int_permutation():
    # the Particle partition applies the same permutation as the int one (mass as the key)
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 7, 64, 1001):
        a = rng.standard_normal((n, 8)).astype(np.float32)
        mass = rng.integers(0, 3, size=n).astype(np.float32) * (rng.random(n) < 0.6)
        a[:, 6] = mass
        tags = np.arange(1, n + 1, dtype=np.int32) * (mass > 0)
        a[:, 7] = np.arange(n)  # radius column doubles as an id
        part, m = ob.partition(a)
        ints, mi = ob.partition_ints(tags)
        assert m == mi == int((mass > 0).sum())
        assert np.all(part[:m, 6] > 0) and np.all(part[m:, 6] <= 0)
        # massive slots hold the same original ids as the int run
        assert [int(t) - 1 for t in ints[:m]] == part[:m, 7].astype(int).tolist()


def test_partition_negative_mass_counts_as_massless():
    a = np.zeros((4, 8), dtype=np.float32)
    a[:, 6] = [-1.0, 2.0, 0.0, 3.0]
    part, m = ob.partition(a)
    assert m == 2 and sorted(part[:2, 6].tolist()) == [2.0, 3.0]


@pytest.mark.parametrize("n", [4096, 1024, 333])
def test_partitioned_ic_digest(manifest, golden, n):
    e = manifest["sets"][str(n)]
    ic = golden(f"ic_{n}.bin")
    assert ob.sha256(ic) == e["ic_sha256"]
    part, m = ob.partition(ic)
    assert m == e["mass_len"]
    assert ob.sha256(part) == e["partitioned_sha256"]


def test_survey_digests_4096(manifest, golden):
    # digests of the reference's GetWorldParticles output
    d = manifest["survey_digests"]
    part, m = ob.partition(golden("ic_4096.bin"))
    assert ob.sha256(part) == d["4096_partitioned"]
    assert ob.sha256(ob.step(part, m, 0.01, 1)) == d["4096_s1_dt0.01"]
    assert ob.sha256(ob.step(part, m, 0.01, 10)) == d["4096_s10_dt0.01"]
    assert ob.sha256(ob.step(part, m, 1.0, 100, kind="avx")) == d["4096_s100_dt1"]


@pytest.mark.parametrize("kind", ["avx_order", "avx"])
@pytest.mark.parametrize("n", [4096, 1024, 333])
def test_bit_exact_against_reference_fixtures(manifest, golden, n, kind):
    e = manifest["sets"][str(n)]
    part, m = ob.partition(golden(f"ic_{n}.bin"))
    for tag, s in e["steps"].items():
        if "file" not in s:
            continue
        want = golden(s["file"])
        got = ob.step(part, m, s["dt"], s["n_steps"], kind=kind)
        assert got.tobytes() == want.tobytes(), f"{kind} differs from reference at N={n} {tag}"


def test_seq_and_f64_agree_with_avx_order_within_fp32(golden, manifest):
    part, m = ob.partition(golden("ic_1024.bin"))
    a = ob.step(part, m, 0.01, 1, kind="avx_order")
    s = ob.step(part, m, 0.01, 1, kind="seq")
    acc64, mag = ob.acc_f64(part, m)
    bound = 1e-4 * np.abs(acc64) + 1e-6 * mag
    assert np.all(np.abs(a[:, 4:6] - acc64) <= bound)
    assert np.all(np.abs(s[:, 4:6] - acc64) <= bound)
    assert not np.array_equal(a[:, 4:6], s[:, 4:6])  # different summation order, different bits


def test_zero_steps_and_empty_sources():
    rng = np.random.default_rng(1)
    a = rng.standard_normal((10, 8)).astype(np.float32)
    a[:, 6] = 0.0
    a[:, 7] = 1.0
    for kind in ("avx_order", "avx", "seq"):
        assert np.array_equal(ob.step(a, 0, 0.5, 0, kind=kind), a)
        out = ob.step(a, 0, 0.5, 2, kind=kind)
        # no sources: acc = 0, straight-line motion
        assert np.all(out[:, 4:6] == 0)
        want = a[:, 0:2] + np.float32(0.5) * a[:, 2:4]
        want = want + np.float32(0.5) * a[:, 2:4]
        assert np.array_equal(out[:, 0:2], want)


def test_thread_count_independent(golden):
    import os
    part, m = ob.partition(golden("ic_1024.bin"))
    a = ob.step(part, m, 0.01, 3, kind="avx")
    sec, used, chk = ob.time_avx_sample(part, m, 0, 256, threads=1)
    assert used == 1 and sec > 0 and np.isfinite(chk)
    b = ob.step(part, m, 0.01, 3, kind="avx")
    assert a.tobytes() == b.tobytes()


def test_sse_and_scalar_builds_of_the_reference(manifest, golden):
    # digests taken from the reference's sim_cpu.c compiled with -DUSE_SSE and with no SIMD define
    v = manifest["simd_variants"]
    assert {e["lanes"] for e in v.values()} == {1, 4}
    for tag, e in v.items():
        part, m = ob.partition(golden(f"ic_{e['n']}.bin"))
        got = ob.step_lanes(part, m, e["dt"], e["n_steps"], e["lanes"])
        assert ob.sha256(got) == e["sha256"], tag
    part, m = ob.partition(golden("ic_333.bin"))
    assert ob.step_lanes(part, m, 0.01, 2, 8).tobytes() == ob.step(part, m, 0.01, 2).tobytes()
    assert ob.step_lanes(part, m, 0.01, 2, 1).tobytes() == ob.step(part, m, 0.01, 2, kind="seq").tobytes()
