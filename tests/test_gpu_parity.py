"""Parity of the HIP path with the oracle, through the C-ABI (needs an MI355X: `pytest -m gpu`).

Tolerance (DESIGN.md "Tolerance"): one step from an identical state, per component,
    |acc_gpu - acc_f64| <= 1e-4 * |acc_f64| + 1e-6 * sum_j |contribution_j|
where acc_f64 is the oracle's float64 evaluation of the same sum.  The reference's own AVX / SSE / scalar
builds differ from each other by the same order (2.9e-6 .. 1.4e-5 relative).  The integrator is
then exact fp32 arithmetic on that acc with the reference's roundings (sim_cpu.c:191-193: mul, then add),
checked bit for bit: vel == vel0 + acc*dt and pos == pos0 + vel*dt.
Ten steps at dt = 0.01: relative L2 over all positions <= 1e-6 AND, the sharp one, relative to what the steps moved:
|(pos - pos0)_gpu - (pos - pos0)_ref| / |(pos - pos0)_ref| <= 1e-4 (rel_displacement; the reference's own sequential
and AVX orders differ by 8.7e-7 there).  Nothing is asserted on 100-step trajectories (chaotic).  Integer-like facts (partition order, mass, radius, pass-through) are bit-exact.
"""
import ctypes as C  # noqa: F401
import os  # noqa: F401
import subprocess  # noqa: F401
import sys  # noqa: F401
import time  # noqa: F401

import numpy as np  # noqa: F401
import pytest

import nbody_amd as nb  # noqa: F401
import oracle_binding as ob  # noqa: F401
from gpu_common import *  # noqa: F401,F403  -- helpers shared by the GPU test files (tests/gpu_common.py)

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------------------------
# fixtures of the reference (golden vectors)
# ---------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("k,w", SHAPES)
@pytest.mark.parametrize("n", [4096, 333])
def test_one_step_against_reference_fixture(golden, manifest, n, k, w, variant):
    part, m = ob.partition(golden(f"ic_{n}.bin"))
    e = manifest["sets"][str(n)]["steps"]
    got = run(part, m, 1, 0.01, variant=variant, k=k, w=w)
    want = golden(e["s1_dt0.01"]["file"]) if "s1_dt0.01" in e else None
    check_one_step(got, part, m, 0.01, want)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("n", [4096, 1024])
def test_ten_steps_against_reference_fixture(golden, manifest, n, variant):
    part, m = ob.partition(golden(f"ic_{n}.bin"))
    want = golden(manifest["sets"][str(n)]["steps"]["s10_dt0.01"]["file"]).astype(np.float64)
    got = run(part, m, 10, 0.01, variant=variant).astype(np.float64)
    rel = np.linalg.norm(got[:, 0:2] - want[:, 0:2]) / np.linalg.norm(want[:, 0:2])
    assert rel <= 1e-6, rel
    relv = np.linalg.norm(got[:, 2:4] - want[:, 2:4]) / np.linalg.norm(want[:, 2:4])
    assert relv <= 1e-5, relv
    # the sharp one: relative to the ten steps' displacement
    assert rel_displacement(got, want, part) <= DISPLACEMENT_TOL, rel_displacement(got, want, part)
    # and the metric does see the physics: a run with all masses zeroed (straight lines) fails it by a wide margin
    straight = part[:, 0:2].astype(np.float64) + 10 * 0.01 * part[:, 2:4].astype(np.float64)
    fake = got.copy()
    fake[:, 0:2] = straight
    assert rel_displacement(fake, want, part) > 10 * DISPLACEMENT_TOL


def test_three_steps_dt005_fixture(golden, manifest):
    part, m = ob.partition(golden("ic_333.bin"))
    want = golden(manifest["sets"]["333"]["steps"]["s3_dt0.05"]["file"]).astype(np.float64)
    got = run(part, m, 3, 0.05).astype(np.float64)
    assert np.linalg.norm(got[:, 0:2] - want[:, 0:2]) / np.linalg.norm(want[:, 0:2]) <= 1e-6
    assert rel_displacement(got, want, part) <= DISPLACEMENT_TOL


# ---------------------------------------------------------------------------------------------------------------
# the World surface and the coherence protocol (reference world.c:76-118, main.c:112-163 call pattern)
# ---------------------------------------------------------------------------------------------------------------

def test_world_gpu_equals_seam(golden):
    ic = golden("ic_1024.bin")
    part, m = ob.partition(ic)
    w = nb.World(ic)
    assert np.array_equal(w.particles(), part)
    w.update_gpu(0.01, 3)
    a = w.particles()
    w.close()
    assert a.tobytes() == run(part, m, 3, 0.01).tobytes()


def test_world_gpu_zero_steps_is_a_noop(golden):
    w = nb.World(golden("ic_333.bin"))
    before = w.particles()
    w.update_gpu(0.01, 0)      # reference world.c:113
    assert np.array_equal(w.particles(), before)
    w.close()


def test_mixed_cpu_gpu_calls_keep_one_state(golden):
    ic = golden("ic_1024.bin")
    part, m = ob.partition(ic)
    w = nb.World(ic)
    w.update_gpu(0.01, 2)
    g2 = w.particles()                       # D2H on demand
    w.update_cpu(0.01, 0)                    # n == 0 still syncs and dirties (world.c:100,109)
    assert np.array_equal(w.particles(), g2)
    w.update_cpu(0.01, 1)                    # CPU continues from the device state
    c3 = w.particles()
    assert np.array_equal(c3, ob.step(g2, m, 0.01, 1))   # CPU path is bit-exact from whatever state it gets
    w.update_gpu(0.02, 1)                    # H2D of the CPU-dirtied array, new dt
    g4 = w.particles()
    w.close()
    check_one_step(g4, c3, m, 0.02)
    check_one_step(run(part, m, 1, 0.01), part, m, 0.01)


def test_read_back_every_frame_equals_one_call(golden):
    part, m = ob.partition(golden("ic_333.bin"))
    sim = nb.SimPipeline(333, m)
    sim.set_data(part)
    for _ in range(5):
        sim.update(1, 0.01)
        sim.get_data()
    a = sim.get_data()
    sim.close()
    assert a.tobytes() == run(part, m, 5, 0.01).tobytes()


@pytest.mark.parametrize("readback", [0, 1, 2])
def test_frame_loop_read_back_modes_give_the_same_frames(golden, readback):
    """The GUI's pattern (reference src/main.c:157-163,237): UpdateWorld_GPU then GetWorldParticles every frame.  With
    readback = 1 / 2 (auto) the merge kernel rides in the update's submission and stores straight into the World's
    page-locked array; every frame must equal the lazy path's, also when the pattern breaks (two updates in a row,
    a CPU step in between, a Get into a foreign buffer through the seam).  The mode is set through the tuning hook on the
    World's own pipeline (the library that ships reads no NB_HIP_READBACK), and read back: a knob that is silently ignored
    fails here instead of comparing the default with itself."""
    ic = golden("ic_1024.bin")
    frames = {}
    for mode in (0, readback):
        w = nb.World(ic)
        assert w.tune(readback=mode) == {"readback": 2}      # the shipped default is auto
        assert w.tune(readback=mode) == {"readback": mode}    # ... and the hook took
        out = []
        for f in range(5):
            w.update_gpu(0.01, 1 + (f & 1))
            out.append(w.particles())
        w.update_gpu(0.01, 1)
        w.update_gpu(0.01, 2)          # update after update: the streak ends, next Get is lazy again
        out.append(w.particles())
        w.update_cpu(0.01, 1)          # pulls (nothing stale), steps on the CPU, marks the host newer
        w.update_gpu(0.01, 1)          # re-upload, step
        out.append(w.particles())
        for f in range(3):
            w.update_gpu(0.005, 3)
            out.append(w.particles())
        assert w.tune(readback=2) == {"readback": mode}       # nothing reset it on the way
        w.close()
        frames[mode] = out
    for a, b in zip(frames[0], frames[readback]):
        assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("n", [1024, 4096])
def test_zero_copy_upload_equals_the_dma_upload(golden, n):
    """SetSimulationData from the World's page-locked array: the split kernel reads the records over PCIe itself
    (default) or after a DMA copy into device staging (tuning hook zero_copy_upload = 0) -- same bytes either way, also
    when the CPU stepper dirtied the array in between (reference world.c:76-81,99-118 protocol)."""
    ic = golden(f"ic_{n}.bin")
    outs = []
    for mode in (1, 0):
        w = nb.World(ic)
        assert w.tune(zero_copy_upload=mode) == {"zero_copy_upload": 1}     # default: zero-copy
        assert w.tune(zero_copy_upload=mode) == {"zero_copy_upload": mode}
        w.update_gpu(0.01, 2)
        w.update_cpu(0.01, 1)
        w.update_gpu(0.01, 1)       # re-upload of the CPU-stepped array
        a = w.particles()
        w.update_cpu(0.01, 0)
        w.update_gpu(0.01, 3)
        outs.append((a, w.particles()))
        w.close()
    assert outs[0][0].tobytes() == outs[1][0].tobytes() and outs[0][1].tobytes() == outs[1][1].tobytes()


def test_eager_read_back_through_the_seam_with_a_foreign_buffer(golden):
    import ctypes as C
    part, m = ob.partition(golden("ic_333.bin"))
    home = part.copy()                                   # the array the pipeline is told about
    sim = nb.SimPipeline(333, m)
    nb.hip_lib().nb_hip_note_host_array(sim._h, home.ctypes.data, home.nbytes)
    sim.configure(readback=1)
    sim.set_data(part)
    sim.update(2, 0.01)                                  # eager: `home` now holds the state
    want = run(part, m, 2, 0.01)
    assert home.tobytes() == want.tobytes()
    assert sim.get_data().tobytes() == want.tobytes()    # Get into another buffer: host copy of the noted array
    sim.step_async(1, 0.01)                              # async steps never write the host array
    sim.sync()
    assert home.tobytes() == want.tobytes()
    assert sim.get_data().tobytes() == run(part, m, 3, 0.01).tobytes()
    nb.hip_lib().nb_hip_note_host_array(sim._h, None, 0)
    sim.close()


# ---------------------------------------------------------------------------------------------------------------
# kernel properties: determinism, variants, linearity, edge shapes
# ---------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("k,w", [(1, 1), (2, 4), (2, 16)])
def test_lds_and_smem_variants_agree_bitwise(golden, k, w):
    # same source order per wave slice; the LDS tail tile's zero-mass pads add exact zeros
    part, m = ob.partition(golden("ic_4096.bin"))
    a = run(part, m, 2, 0.01, variant=0, k=k, w=w)
    b = run(part, m, 2, 0.01, variant=1, k=k, w=w, unit=64)   # the LDS route always slices by whole 64-source tiles
    assert a.tobytes() == b.tobytes()
    assert a.tobytes() == run(part, m, 2, 0.01, variant=0, k=k, w=w, unit=8).tobytes()   # asked for 8: still 64 there


def test_source_count_sweep_both_routes_agree_bitwise():
    """The scalar-cache route walks its slice in pairs of 8-source groups inside 256-source blocks with a ragged tail;
    the LDS route walks 64-source tiles.  Same arithmetic, same order: every source count around those boundaries
    must give the same bits on both routes, for long slices (W = 1), short ones (W = 16) and split/passes."""
    counts = sorted(set(list(range(1, 40)) + [63, 64, 65, 71, 72, 73, 127, 128, 129, 247, 248, 249, 255, 256, 257, 263, 264,
                                              265, 511, 512, 513, 519, 520, 1023, 1024, 1025, 1031, 2047, 2048, 2049, 2111]))
    for m_want in counts:
        n = m_want + 37
        part, m = synth(n, 1.0, seed=m_want)
        part[m_want:, 6] = 0.0                       # exactly m_want sources, 37 massless receivers
        part, m = ob.partition(part)
        assert m == m_want
        for knobs in (dict(k=1, w=1), dict(k=2, w=16), dict(k=2, w=4, split=3), dict(k=1, w=4, passes=2)):
            a = run(part, m, 2, 0.01, variant=0, **knobs)
            b = run(part, m, 2, 0.01, variant=1, unit=64, **knobs)   # same granule as the LDS route's whole tiles
            assert a.tobytes() == b.tobytes(), f"routes differ at {m_want} sources, {knobs}"
        if m_want in (1, 9, 64, 257, 1031):
            check_one_step(run(part, m, 1, 0.01, variant=1, k=1, w=1), part, m, 0.01)


@pytest.mark.parametrize("unit", [8, 16, 32])
def test_fine_source_granules(unit):
    """Latency-bound launches slice the sources in granules of 8 / 16 / 32 instead of 64 (StepParams::unit): every
    source must still be added exactly once whatever the count, the split and the waves per workgroup -- checked against
    float64 at source counts around every granule, tile and block boundary, and against the 64-source granule."""
    counts = [1, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 129, 250, 255, 256, 257, 263, 511, 520, 1000, 1031, 2049]
    for m_want in counts:
        n = m_want + 70
        part, m = synth(n, 1.0, seed=1000 + m_want)
        part[m_want:, 6] = 0.0
        part, m = ob.partition(part)
        assert m == m_want
        coarse = run(part, m, 1, 0.01, k=1, w=1, unit=64)
        for knobs in (dict(k=1, w=16), dict(k=2, w=4, split=3), dict(k=1, w=8, split=16), dict(k=2, w=16, passes=2), dict(k=1, w=1)):
            got = run(part, m, 1, 0.01, unit=unit, **knobs)
            acc64, mag = ob.acc_f64(part, m)
            err = np.abs(got[:, 4:6].astype(np.float64) - acc64)
            assert np.all(err <= acc_bound(acc64, mag)), f"{m_want} sources, unit {unit}, {knobs}: {np.max(err / acc_bound(acc64, mag)):.3f}"
            assert np.all(np.abs(got[:, 4:6].astype(np.float64) - coarse[:, 4:6]) <= 2 * acc_bound(acc64, mag))
            v = part[:, 2:4] + got[:, 4:6] * np.float32(0.01)
            assert np.array_equal(got[:, 2:4], v) and np.array_equal(got[:, 0:2], part[:, 0:2] + v * np.float32(0.01))
        # one wave walking everything in granules of `unit` adds the sources in index order: same bits as the 64 granule
        assert run(part, m, 1, 0.01, k=1, w=1, unit=unit).tobytes() == coarse.tobytes()


def test_auto_shape_uses_fine_granules_on_small_worlds(golden):
    part, m = ob.partition(golden("ic_333.bin"))
    sim = nb.SimPipeline(333, m)
    sim.configure(lanes=1)          # the classic kernel's shape model (the all-auto default is a lane-split launch here)
    sim.set_data(part)
    sim.update(2, 0.01)
    shape = sim.launch_shape()
    got = sim.get_data()
    sim.close()
    assert shape["unit"] in (8, 16) and shape["w"] == 16 and shape["split"] == 1     # 150-odd sources over 16 waves
    want = ob.step(part, m, 0.01, 2)
    assert rel_l2_pos(got, want) <= 1e-6
    lds = nb.SimPipeline(333, m)
    lds.configure(variant=0)
    lds.set_data(part)
    lds.update(1, 0.01)
    assert lds.launch_shape()["unit"] == 64                                             # the LDS route: whole tiles
    lds.close()


@pytest.mark.parametrize("P,n,frac", [(2, 700, 0.9), (3, 1500, 0.6), (8, 5000, 0.8), (5, 333, 1.0)])
def test_overlapped_shards_both_routes_agree_bitwise(P, n, frac):
    # the overlapped step walks two source ranges (own slice, then the rest): the second range starts mid-slice
    part, m = synth(n, frac, seed=n + P)
    outs = []
    for variant in (0, 1):
        g = nb.LocalShardGroup(n, m, P, overlap=1, variant=variant, k=2, w=4, unit=64)   # one granule for both routes
        g.set_data(part)
        g.step(2, 0.01)
        outs.append(g.get_data(P - 1))
        g.close()
    assert outs[0].tobytes() == outs[1].tobytes()
    g = nb.LocalShardGroup(n, m, P, overlap=1)
    g.set_data(part)
    g.step(1, 0.01)
    got = g.get_data(0)
    g.close()
    check_one_step(got, part, m, 0.01)


def test_deterministic(golden):
    part, m = ob.partition(golden("ic_4096.bin"))
    assert run(part, m, 3, 0.01).tobytes() == run(part, m, 3, 0.01).tobytes()


def test_k_does_not_change_bits(golden):
    # register blocking regroups receivers, never the order sources are added in
    part, m = ob.partition(golden("ic_1024.bin"))
    a = run(part, m, 1, 0.01, k=1, w=4, split=1)
    for w in (4, 16):
        assert run(part, m, 1, 0.01, k=2, w=w, split=1).tobytes() == run(part, m, 1, 0.01, k=1, w=w, split=1).tobytes()
    assert run(part, m, 1, 0.01, k=2, w=4, split=1).tobytes() == a.tobytes()


@pytest.mark.parametrize("split", [1, 2, 3, 7, 16])
@pytest.mark.parametrize("variant", [0, 1])
def test_source_split_steps(golden, split, variant):
    # gridDim.y workgroups per receiver tile + the finish kernel; parts are added in part order
    part, m = ob.partition(golden("ic_4096.bin"))
    got = run(part, m, 1, 0.01, split=split, variant=variant)
    check_one_step(got, part, m, 0.01)
    chained = run(part, m, 5, 0.01, split=split, variant=variant, graph=1)
    plain = run(part, m, 5, 0.01, split=split, variant=variant, graph=0)
    assert chained.tobytes() == plain.tobytes()


@pytest.mark.parametrize("n,frac", [(9000, 0.5), (12000, 0.3), (20011, 0.5), (700, 1.0)])
def test_fused_finish_equals_the_two_kernel_form(n, frac):
    """Split steps whose last-arriving workgroup per receiver tile adds the parts inside the step kernel (knob
    "fused_finish": parts as agent-scope stores / loads, one ticket per tile) against the same shape with the finish
    kernel: same part order, same roundings -> the same bits, as plain launches, as a hipGraph chain, with source passes
    chained through acc[], for every (k, w) the route instantiates -- and both are the reference's step (float64 bound,
    integrator bit-exact)."""
    part, m = synth(n, frac, seed=n)
    for knobs in (dict(), dict(k=1, w=4, split=13), dict(k=2, w=8, split=7), dict(k=2, w=16, split=2, passes=2),
                  dict(k=1, w=16, split=16, unit=8), dict(k=1, w=8, split=3, passes=3)):
        two = run(part, m, 1, 0.01, fused_finish=0, lanes=1, **knobs)
        one = run(part, m, 1, 0.01, fused_finish=1, lanes=1, **knobs)
        assert one.tobytes() == two.tobytes(), (n, knobs)
        for graph in (0, 1):
            assert run(part, m, 7, 0.01, fused_finish=1, lanes=1, graph=graph, **knobs).tobytes() == \
                run(part, m, 7, 0.01, fused_finish=0, lanes=1, graph=graph, **knobs).tobytes(), (n, knobs, graph)
    check_one_step(run(part, m, 1, 0.01, fused_finish=1, lanes=1, split=5), part, m, 0.01)


def test_fused_finish_auto_policy():
    """Auto: unsharded split steps on the scalar-cache route from N x M >= 4e7 up to 200 000 receivers run WITHOUT the finish
    kernel; smaller worlds, the LDS-tile route, the BASELINE sizes and sharded steps keep it."""
    def finish_launches(n, **knobs):
        _, part, m = bench_universe(n)
        sim = nb.SimPipeline(n, m)
        sim.configure(**knobs)
        sim.set_data(part)
        sim.update(3, 0.01)
        out = (sim.finish_launches(), sim.launch_shape()["split"], nb.plan_launch(n, m)["fused_finish"])
        sim.close()
        return out
    f, split, planned = finish_launches(10000)
    assert split > 1 and f == 0 and planned == 1                     # one kernel per step
    f, split, planned = finish_launches(10000, fused_finish=0)
    assert split > 1 and f == 3
    f, split, planned = finish_launches(6000)
    assert split > 1 and f == 3 and planned == 0                     # below the rule: loses inside a hipGraph
    f, split, planned = finish_launches(6000, fused_finish=1)
    assert split > 1 and f == 0
    f, split, planned = finish_launches(10000, variant=0)
    assert split > 1 and f == 3                                      # the LDS-tile route has no fused instantiation
    f, split, planned = finish_launches(262144)
    assert split > 1 and f > 0 and planned == 0                      # BASELINE sizes: the profiled two-kernel form
    _, part, m = bench_universe(20000)
    g = nb.LocalShardGroup(20000, m, 2, split=4)
    g.set_data(part)
    g.step(1, 0.01)                                                  # sharded steps keep the two-kernel form; parity:
    sharded = g.get_data(0)
    g.close()
    check_one_step(sharded, part, m, 0.01)


def test_source_split_more_parts_than_chunks():
    part, m = synth(300, 0.1, seed=2)       # ~30 sources = one chunk, 16 parts: most parts are empty
    got = run(part, m, 1, 0.02, split=16)
    check_one_step(got, part, m, 0.02)


def test_auto_shape_reports_its_choice(golden):
    part, m = ob.partition(golden("ic_4096.bin"))
    sim = nb.SimPipeline(4096, m)
    sim.set_data(part)
    sim.update(1, 0.01)
    shape = sim.launch_shape()
    sim.close()
    assert shape["k"] in (1, 2) and shape["w"] in (4, 8, 16) and 1 <= shape["split"] <= 16 and shape["lanes"] in (1, 2, 4, 8)
    # classic: tiles of 64 * k receivers x source parts; lane-split: 64 / lanes receivers per workgroup, one part
    assert shape["workgroups"] == -(-4096 * shape["lanes"] // (64 * shape["k"])) * shape["split"]


def test_acc_is_linear_in_mass_by_powers_of_two(golden):
    part, m = ob.partition(golden("ic_1024.bin"))
    heavy = part.copy()
    heavy[:, 6] *= 4.0
    a = run(part, m, 1, 0.01)[:, 4:6]
    b = run(heavy, m, 1, 0.01)[:, 4:6]
    assert np.array_equal(b, a * np.float32(4.0))


@pytest.mark.parametrize("n,frac", [(1, 1.0), (2, 0.5), (63, 0.5), (64, 1.0), (65, 0.3), (130, 1.0), (257, 0.02),
                                    (1000, 0.0), (4097, 0.7)])
@pytest.mark.parametrize("variant", [0, 1])
def test_ragged_sizes_and_source_counts(n, frac, variant):
    part, m = synth(n, frac, seed=n)
    if frac == 1.0:
        assert m == n
    got = run(part, m, 1, 0.02, variant=variant)
    check_one_step(got, part, m, 0.02)


def test_no_sources_means_straight_lines():
    part, m = synth(500, 0.0, seed=9)
    assert m == 0
    got = run(part, m, 2, 0.5)
    want = ob.step(part, 0, 0.5, 2)
    assert got.tobytes() == want.tobytes()


def test_single_source_known_answer():
    # one core at the origin, one tracer at (3, 4): |d|^2 + radius = 25 + 0 ... use radius 11 -> r2 = 36
    a = np.zeros((2, 8), dtype=np.float32)
    a[0, 6], a[0, 7] = 21.6, 1.0           # G*m = 216
    a[1, 0], a[1, 1], a[1, 7] = 3.0, 4.0, 11.0
    got = run(a, 1, 1, 1.0)
    # f = 216 / 36^1.5 = 1 -> acc = -(3, 4)
    assert np.allclose(got[1, 4:6], [-3.0, -4.0], rtol=1e-6, atol=0)
    assert got[0, 4] == 0 and got[0, 5] == 0   # self-interaction contributes exactly zero
    assert np.allclose(got[1, 0:2], [0.0, 0.0], atol=1e-6)


def test_zero_radius_singularities_match_the_reference():
    # reference behaviour (SURVEY.md 8a "edge semantics"): radius == 0 plus a coincident source is not guarded:
    # 0 * inf = NaN.  A massive body with radius 0 meets itself; a tracer with radius 0 sits on a source.
    a = np.zeros((4, 8), dtype=np.float32)
    a[:, 0] = [0.0, 50.0, 50.0, 90.0]
    a[:, 6] = [5.0, 7.0, 0.0, 0.0]
    a[:, 7] = [0.0, 1.0, 0.0, 0.5]          # body 0: radius 0 (self-hit); tracer 2: radius 0 on top of body 1
    part, m = ob.partition(a)
    want = ob.step(part, m, 0.1, 1)
    got = run(part, m, 1, 0.1)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.isnan(got[0, 4]) and np.isnan(got[2, 4]) and not np.isnan(got[1, 4]) and not np.isnan(got[3, 4])
    ok = ~np.isnan(want)
    assert np.allclose(got[ok], want[ok], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("seed", range(8))
def test_random_small_worlds(seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(1, 700))
    part, m = synth(n, float(rng.random()), seed=seed, extent=float(10 ** rng.uniform(1, 5)))
    dt = float(10 ** rng.uniform(-3, -1))
    got = run(part, m, 1, dt, variant=seed & 1, split=int(rng.integers(0, 5)))
    check_one_step(got, part, m, dt)


@pytest.mark.parametrize("block", range(4))
def test_random_worlds_random_knobs(block):
    """Seeded fuzz over world sizes, massive fractions and every launch knob (receivers per lane, waves per workgroup,
    source split, slice granule, source passes, source route, lane groups, one-workgroup chain, graph policy): one step
    must sit within the float64 tolerance with an exact integrator, and a three-step chain must give the same bytes as
    plain launches."""
    rng = np.random.default_rng(9000 + block)
    for case in range(12):
        n = int(rng.choice([1, 2, 63, 64, 65, 127, 300, 777, 1024, 1500, 2111, 3000]))
        frac = float(rng.choice([0.02, 0.3, 0.5, 1.0]))
        part, m = synth(n, frac, seed=int(rng.integers(1 << 30)), extent=float(rng.choice([1e2, 1e4, 1e6])))
        knobs = dict(k=int(rng.choice([0, 1, 2])), w=int(rng.choice([0, 1, 4, 8, 16])), split=int(rng.integers(0, 17)),
                     unit=int(rng.choice([0, 8, 16, 32, 64])), passes=int(rng.choice([0, 1, 2, 3])),
                     variant=int(rng.choice([0, 1])), lanes=int(rng.choice([0, 0, 1, 2, 4, 8])),
                     fused_chain=int(rng.choice([0, 1, 2])))
        dt = float(rng.choice([0.01, 0.005, 0.02]))
        one = run(part, m, 1, dt, **knobs)
        acc64, mag = ob.acc_f64(part, m)
        err = np.abs(one[:, 4:6].astype(np.float64) - acc64)
        assert np.all(err <= acc_bound(acc64, mag)), f"case {block}.{case}: n={n} m={m} {knobs}: {np.max(err / acc_bound(acc64, mag)):.3f}"
        v = part[:, 2:4] + one[:, 4:6] * np.float32(dt)
        assert np.array_equal(one[:, 2:4], v) and np.array_equal(one[:, 0:2], part[:, 0:2] + v * np.float32(dt)), (n, knobs)
        assert np.array_equal(one[:, 6:8], part[:, 6:8])
        assert run(part, m, 3, dt, graph=1, **knobs).tobytes() == run(part, m, 3, dt, graph=0, **knobs).tobytes(), (n, knobs)


def test_receivers_are_independent_of_each_other():
    """A receiver's new state depends on the sources and on itself, never on which other receivers exist (reference
    particle_cs.glsl:32-54: each invocation reads all sources and writes only its own particle).  With one wave walking
    all sources in index order (w = 1) that holds bit for bit: appending massless receivers, or dropping some, leaves
    every other particle's result untouched -- on the classic kernel and, with the slicing fixed by the source count
    alone, on the lane-split one."""
    part, m = synth(900, 0.4, seed=77)
    extra, _ = synth(300, 0.0, seed=78)             # massless only
    bigger = np.concatenate([part, extra])
    fewer = part[: m + 100]
    for knobs in (dict(k=1, w=1), dict(k=2, w=1), dict(lanes=4, w=8), dict(lanes=8, w=16)):
        base = run(part, m, 2, 0.01, **knobs)
        more = run(bigger, m, 2, 0.01, **knobs)
        less = run(fewer, m, 2, 0.01, **knobs)
        assert more[:900].tobytes() == base.tobytes(), knobs
        assert less.tobytes() == base[: m + 100].tobytes(), knobs


def test_negative_mass_is_massless():
    a = np.zeros((3, 8), dtype=np.float32)
    a[:, 7] = 1.0
    a[:, 6] = [5.0, -3.0, 0.0]
    a[:, 0] = [0.0, 10.0, 20.0]
    w = nb.World(a)                          # partition uses > 0 / <= 0 (reference world.c:35-36)
    w.update_gpu(0.1, 1)
    got = w.particles()
    w.close()
    assert got[0, 6] == 5.0 and got[0, 4] == 0.0
    assert np.all(got[1:, 4] < 0)            # both tracers fall towards the single source, nothing else pulls


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json sizes: spot-checked against float64 on a receiver sample + size-independent properties
# ---------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("variant", [1, 0], ids=["scalar-cache", "lds-tiles"])
@pytest.mark.parametrize("n,steps,dt", [(65536, 1, 0.01), (262144, 4, 0.005), (1 << 20, 1, 0.01)])
def test_baseline_sizes_spot_check(n, steps, dt, variant):
    """Both source routes -- the default scalar-cache one and the north star's LDS-staged tiles -- against float64 at
    every single-GPU BASELINE size."""
    ic = nb.make_galaxies(n, 2, seed=11037)          # the bench's universe
    w = nb.World(ic)
    part = w.particles()
    m = int((part[:, 6] > 0).sum())
    w.close()
    sim = nb.SimPipeline(n, m)
    sim.configure(graph=1, variant=variant)           # chains as hipGraphs from their first use (config 3)
    sim.set_data(part)
    sim.update(1, dt)
    one = sim.get_data()
    assert sim.launch_shape()["variant"] == ("smem" if variant else "lds")
    rng = np.random.default_rng(n)
    idx = np.unique(np.concatenate([[0, 1, m - 1, m, n - 1], rng.integers(0, n, 500)])).astype(np.uint32)
    acc64, mag = ob.acc_f64_subset(part, m, idx)
    err = np.abs(one[idx, 4:6].astype(np.float64) - acc64)
    bound = acc_bound(acc64, mag)
    assert np.all(err <= bound), f"worst ratio {np.max(err / bound):.3f}"
    # tie-breaker: the device sum is closer to float64 than the reference's AVX sum is
    e_avx = np.abs(ob.acc_avx_subset(part, m, idx).astype(np.float64) - acc64)
    assert np.sqrt(np.mean((err / mag) ** 2)) <= np.sqrt(np.mean((e_avx / mag) ** 2))
    # against the AVX order itself: a stated constant per size (GPU_VS_AVX below), not the triangle inequality
    assert np.all(np.abs(one[idx, 4:6].astype(np.float64) - ob.acc_avx_subset(part, m, idx)) <= GPU_VS_AVX[n][0] * mag)
    # integrator exactness on the device's own acc (mul, add roundings of the reference)
    v = part[:, 2:4] + one[:, 4:6] * np.float32(dt)
    p = part[:, 0:2] + v * np.float32(dt)
    assert np.array_equal(one[:, 2:4], v) and np.array_equal(one[:, 0:2], p)
    if steps > 1:
        sim.update(steps - 1, dt)                     # multi-step hipGraph chain (config 3)
        chained = sim.get_data()
        sim.set_data(part)
        sim.configure(graph=0)
        sim.update(steps, dt)
        assert sim.get_data().tobytes() == chained.tobytes()
    sim.close()


def test_ten_steps_at_config2_size_against_the_avx_path():
    """BASELINE config 2 (N = 65 536), TEN steps at dt = 0.01 -- the reference semantics world.c:99-110 x 10 -- against
    the bit-exact restatement of the reference's AVX stepper (sim_cpu.c:156-194), single pipeline and through P = 8
    shards with the gather in-stream and overlapped.  Tolerance: relative L2 over all positions <= 1e-6 (the same
    bound as the 4 096-particle fixtures; velocities <= 1e-4: they carry the summation-order difference undamped)."""
    n = 65536
    _, part, m = bench_universe(n)
    want = ob.step(part, m, 0.01, 10, kind="avx")
    got = run(part, m, 10, 0.01)
    assert rel_l2_pos(got, want) <= 1e-6
    assert rel_displacement(got, want, part) <= DISPLACEMENT_TOL, rel_displacement(got, want, part)
    dv = got[:, 2:4].astype(np.float64) - want[:, 2:4]
    assert np.linalg.norm(dv) / np.linalg.norm(want[:, 2:4].astype(np.float64)) <= 1e-4
    assert np.array_equal(got[:, 6:8], want[:, 6:8])
    for overlap in (0, 1):
        g = nb.LocalShardGroup(n, m, 8, overlap=overlap)
        g.set_data(part)
        g.step(10, 0.01)
        sharded = g.get_data(overlap * 7)
        g.close()
        assert rel_l2_pos(sharded, want) <= 1e-6, f"P=8 overlap={overlap}"
        assert rel_l2_pos(sharded, got) <= 1e-6
        assert rel_displacement(sharded, want, part) <= DISPLACEMENT_TOL, f"P=8 overlap={overlap}"
        assert np.array_equal(sharded[:, 6:8], want[:, 6:8])


@pytest.mark.parametrize("n", sorted(GPU_VS_AVX))
def test_gpu_versus_the_avx_path_at_every_baseline_size(n):
    """north_star's parity claim as numbers (reference src/lib/sim_cpu.c:156-194, world.c:99-110): at N = 65 536 /
    262 144 / 2^20 the GPU step against the reference AVX order itself -- not only each against float64 -- one step on a
    sample, and ten (two at 2^20) full steps of the AVX stepper on the host cores against the same steps on the GPU."""
    sys.path.insert(0, os.path.join(nb.ROOT, "tools"))
    import gpu_vs_avx
    bound, steps = GPU_VS_AVX[n]
    r = gpu_vs_avx.measure(n, steps)
    print(gpu_vs_avx.line(r))
    assert r["gpu_avx_max"] <= bound, r
    assert r["gpu_f64_max"] <= r["avx_f64_max"]              # the difference is carried by the AVX order's error
    # ... at the first step and at every later one, each from the GPU's own state (float64 is the tie-breaker)
    assert r["later_steps_tie_break_holds"] and r["later_steps_gpu_f64_max"] <= r["later_steps_avx_f64_min"], r
    assert r["rel_displacement"] <= 5e-5, r                   # measured <= 1.5e-5; the stated multi-step tolerance is 1e-4
    assert r["rel_l2_vel"] <= 1e-4 and r["rel_l2_pos"] <= 1e-6 and r["static_equal"]


def test_config3_dt_halved_on_a_cached_chain():
    """BASELINE config 3 literally: N = 262 144, a multi-step hipGraph chain, then the SAME cached chain at dt/2 and
    back.  The step size lives in device memory like the reference's uniform block, so halving it is the reference's
    own mechanism -- a small in-stream write when dt differs from the cached one (sim_gpu.c:268-284) -- and the
    instantiated chain is replayed untouched."""
    n = 262144
    _, part, m = bench_universe(n)
    sim = nb.SimPipeline(n, m)
    sim.configure(graph=1)
    sim.set_data(part)
    sim.update(4, 0.01)
    assert sim.graph_stats() == {"cached": 1, "dt_uploads": 1}
    sim.update(4, 0.005)            # cached chain, dt halved
    assert sim.graph_stats() == {"cached": 1, "dt_uploads": 2}
    sim.update(4, 0.005)            # same dt: nothing written
    sim.update(4, 0.01)
    assert sim.graph_stats() == {"cached": 1, "dt_uploads": 3}
    got = sim.get_data()
    sim.close()
    ref = nb.SimPipeline(n, m)
    ref.configure(graph=0)
    ref.set_data(part)
    for dt in (0.01, 0.005, 0.005, 0.01):
        ref.update(4, dt)
    want = ref.get_data()
    ref.close()
    assert got.tobytes() == want.tobytes()
    # and the halved-dt step itself is right: one step at dt/2 from the initial state, float64 spot check
    one = run(part, m, 1, 0.005, graph=1)
    idx = np.unique(np.random.default_rng(3).integers(0, n, 300)).astype(np.uint32)
    acc64, mag = ob.acc_f64_subset(part, m, idx)
    assert np.all(np.abs(one[idx, 4:6].astype(np.float64) - acc64) <= acc_bound(acc64, mag))
    v = part[:, 2:4] + one[:, 4:6] * np.float32(0.005)
    assert np.array_equal(one[:, 2:4], v) and np.array_equal(one[:, 0:2], part[:, 0:2] + v * np.float32(0.005))


def test_odd_chain_lengths_get_one_cached_graph_per_phase(golden):
    """A frame loop that asks for the same odd n alternates between the two ping-pong phases: the cache holds one
    instantiated chain per phase and replays them untouched (no node is re-patched), and stays bounded."""
    part, m = ob.partition(golden("ic_333.bin"))
    sim = nb.SimPipeline(333, m)
    sim.configure(graph=1)
    sim.set_data(part)
    for _ in range(6):
        sim.update(3, 0.01)
    assert sim.graph_stats() == {"cached": 2, "dt_uploads": 1}
    for n in range(1, 30):          # many chain lengths: least recently used chains are evicted
        sim.update(n, 0.01)
    assert sim.graph_stats()["cached"] <= 8
    got = sim.get_data()
    sim.close()
    assert got.tobytes() == run(part, m, 18 + sum(range(1, 30)), 0.01, graph=0).tobytes()


def test_world_level_path_at_baseline_size():
    """The include/nbody.h surface at N = 2^20 (reference world.c:76-118 semantics): CreateWorld -> UpdateWorld_GPU(2)
    -> GetWorldParticles -> UpdateWorld_CPU(0) (syncs from the GPU and dirties the array, world.c:100,109) ->
    UpdateWorld_GPU(1) (re-upload of the 32 MiB array the World page-locked).  Bytes must equal the bare seam fed
    the same way."""
    n = 1 << 20
    ic, part, m = bench_universe(n)
    w = nb.World(ic)
    assert np.array_equal(w.particles(), part)
    w.update_gpu(0.01, 2)
    two = w.particles()
    w.update_cpu(0.01, 0)
    w.update_gpu(0.01, 1)
    three = w.particles()
    frames = []
    for _ in range(3):              # a frame loop: from the third update -> Get pair on, the read-back is eager (32 MiB of
        w.update_gpu(0.01, 1)       # kernel stores straight into the page-locked array inside the update's submission)
        frames.append(w.particles())
    w.close()
    sim = nb.SimPipeline(n, m)
    sim.set_data(part)
    sim.update(2, 0.01)
    want2 = sim.get_data()
    sim.set_data(want2)             # what the World's re-upload after UpdateWorld_CPU(0) amounts to
    sim.update(1, 0.01)
    want3 = sim.get_data()
    want_frames = []
    for _ in range(3):
        sim.update(1, 0.01)
        want_frames.append(sim.get_data())
    sim.close()
    assert two.tobytes() == want2.tobytes()
    assert three.tobytes() == want3.tobytes()
    for got, want in zip(frames, want_frames):
        assert got.tobytes() == want.tobytes()
    # and the state is the right one: spot check of the third step against float64
    idx = np.unique(np.random.default_rng(11).integers(0, n, 200)).astype(np.uint32)
    acc64, mag = ob.acc_f64_subset(two, m, idx)
    assert np.all(np.abs(three[idx, 4:6].astype(np.float64) - acc64) <= acc_bound(acc64, mag))
