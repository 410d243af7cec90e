"""Step chains on the GPU (needs an MI355X: `pytest -m gpu`): hipGraph vs plain launches, phases, dt changes on a cached chain, the
one-workgroup chain and the lane-split launches of small worlds -- every schedule anchored to the reference AVX stepper and to float64
segment by segment (tolerances: tests/test_gpu_parity.py docstring; helpers: tests/gpu_common.py).
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
# step chains: hipGraph vs plain launches, phases, dt patching
# ---------------------------------------------------------------------------------------------------------------



@pytest.mark.parametrize("n_steps", [1, 2, 3, 7, 64, 65, 130])
def test_graph_chain_equals_plain_launches(golden, n_steps):
    part, m = ob.partition(golden("ic_333.bin"))
    a = run(part, m, n_steps, 0.01, graph=1)
    b = run(part, m, n_steps, 0.01, graph=0)
    assert a.tobytes() == b.tobytes()
    # ... and both are the reference's trajectory, not merely each other's
    assert_anchored(a, part, m, [(n_steps, 0.01)], "graph chain")


@pytest.mark.parametrize("graph", [1, 2])
def test_split_calls_and_odd_phases(golden, graph):
    # graph = 2 (the default) on a small world: one canonical 32-step chain, prebuilt at set_data, replayed by calls of
    # 16+ steps (plain launches to reach phase 0 and for the remainder); shorter calls are plain launches
    part, m = ob.partition(golden("ic_333.bin"))
    calls = (3, 3, 1, 5, 3, 3, 17, 17, 16, 17, 17, 40, 33, 71, 32, 1, 64)   # lengths reused on the other ping-pong phase
    want = run(part, m, sum(calls), 0.01, graph=0)
    sim = nb.SimPipeline(333, m)
    sim.configure(graph=graph)
    sim.set_data(part)
    for n in calls:
        sim.update(n, 0.01)
    got = sim.get_data()
    stats = sim.graph_stats()
    sim.close()
    assert got.tobytes() == want.tobytes()
    assert stats["cached"] == (8 if graph == 1 else 1)   # always: one per (length, phase), capped at 8; auto: the canonical one
    assert stats["dt_uploads"] == 1
    # the WHOLE 343-step schedule, call by call, against the reference stepper with float64 as the tie-break -- and its
    # first calls (7 steps) at the stated tolerance
    assert_anchored(got, part, m, [(n, 0.01) for n in calls], f"split calls graph={graph}")
    head = nb.SimPipeline(333, m)
    head.configure(graph=graph)
    head.set_data(part)
    for n in calls[:3]:
        head.update(n, 0.01)
    early = head.get_data()
    head.close()
    assert_anchored(early, part, m, [(n, 0.01) for n in calls[:3]], "split calls, first three")


@pytest.mark.parametrize("graph", [1, 2])
def test_dt_change_patches_the_cached_chain(golden, graph):
    part, m = ob.partition(golden("ic_333.bin"))
    n = 4 if graph == 1 else 40  # auto mode: 40 = one replay of the canonical 32-step chain + 8 plain launches
    sim = nb.SimPipeline(333, m)
    sim.configure(graph=graph)
    sim.set_data(part)
    sim.update(n, 0.01)
    sim.update(n, 0.005)         # same n, dt halved: a 4-byte write to device memory, the cached chain is untouched
    sim.update(n, 0.01)
    sim.update(n, 0.0025)
    got = sim.get_data()
    assert sim.graph_stats() == {"cached": 1, "dt_uploads": 4}
    sim.close()
    ref = nb.SimPipeline(333, m)
    ref.configure(graph=0)
    ref.set_data(part)
    for dt in (0.01, 0.005, 0.01, 0.0025):
        ref.update(n, dt)
    want = ref.get_data()
    ref.close()
    assert got.tobytes() == want.tobytes()
    # the dt the chain read from device memory is the dt the reference path was given, call by call
    assert_anchored(got, part, m, [(n, dt) for dt in (0.01, 0.005, 0.01, 0.0025)], f"dt change graph={graph}")
    wrong = avx_steps(part, m, [(n, 0.01)] * 4)          # had the chain kept its first dt, it would be here
    assert rel_displacement(got, wrong, part) > 0.1


def test_long_runs_replay_the_canonical_chain_exactly(golden):
    """5 000 steps in auto mode (the prebuilt 32-step chain replayed 156 times + 8 plain launches) = 50 calls of 100
    steps = 5 000 plain launches, bit for bit; the pipeline keeps one cached chain throughout."""
    part, m = ob.partition(golden("ic_1024.bin"))
    want = run(part, m, 5000, 0.001, graph=0)
    assert run(part, m, 5000, 0.001).tobytes() == want.tobytes()
    sim = nb.SimPipeline(1024, m)
    sim.set_data(part)
    for _ in range(50):
        sim.update(100, 0.001)
    stats = sim.graph_stats()
    got = sim.get_data()
    sim.close()
    assert got.tobytes() == want.tobytes() and stats == {"cached": 1, "dt_uploads": 1}
    assert np.all(np.isfinite(got))


def test_set_data_again_restarts_from_the_new_state(golden):
    part, m = ob.partition(golden("ic_333.bin"))
    sim = nb.SimPipeline(333, m)
    sim.set_data(part)
    sim.update(3, 0.01)
    sim.set_data(part)
    sim.update(2, 0.01)
    got = sim.get_data()
    sim.close()
    assert got.tobytes() == run(part, m, 2, 0.01).tobytes()
    want = avx_steps(part, m, [(2, 0.01)])
    assert rel_displacement(got, want, part) <= DISPLACEMENT_TOL


def test_async_steps_then_sync(golden):
    part, m = ob.partition(golden("ic_333.bin"))
    sim = nb.SimPipeline(333, m)
    sim.set_data(part)
    sim.step_async(2, 0.01)
    sim.step_async(3, 0.01)
    sim.sync()
    ms, launches = sim.last_step_ms()
    got = sim.get_data()
    sim.close()
    assert launches == 3 and ms > 0
    assert got.tobytes() == run(part, m, 5, 0.01).tobytes()
    assert rel_displacement(got, avx_steps(part, m, [(5, 0.01)]), part) <= DISPLACEMENT_TOL


# ---------------------------------------------------------------------------------------------------------------
# the one-workgroup chain (knob "fused_chain"): a whole n-step call inside ONE launch, positions in LDS
# ---------------------------------------------------------------------------------------------------------------



@pytest.mark.parametrize("n_steps", [2, 3, 7, 64, 65, 130])
@pytest.mark.parametrize("n", [100, 250, 333, 512])
def test_fused_chain_equals_plain_launches(n, n_steps):
    """n steps inside one launch == n per-step launches of the matching shape (k = 2, w = 16 / tiles, split = 1,
    unit = 8), bit for bit: same interaction statements, same source slices, same reduction order, same integrator
    roundings; only the kernel boundaries are gone."""
    part, m = bench_universe(n)[1:] if n >= 200 else synth(n, 0.5, seed=n)   # MakeGalaxies needs 100 per galaxy
    sim = nb.SimPipeline(n, m)
    sim.configure(fused_chain=1)
    sim.set_data(part)
    sim.update(n_steps, 0.01)
    assert sim.fused_steps() == n_steps and sim.last_step_ms()[1] == 1          # ONE launch
    shape = sim.launch_shape()
    assert {key: shape[key] for key in ("k", "w", "split", "unit")} == matched_shape(n) and shape["workgroups"] == 1
    got = sim.get_data()
    sim.close()
    want = run(part, m, n_steps, 0.01, fused_chain=0, graph=0, **matched_shape(n))
    assert got.tobytes() == want.tobytes()
    assert got.tobytes() == run(part, m, n_steps, 0.01, fused_chain=0, graph=1, **matched_shape(n)).tobytes()
    assert_anchored(got, part, m, [(n_steps, 0.01)], f"fused chain N={n}", fused_chain=0, **matched_shape(n))


def test_fused_chain_auto_policy_and_split_calls():
    """Auto: calls of 2+ steps on worlds with N <= 256 and N x M <= 3.6e4 (the reference harness' N = 250 row) run fused, single steps
    and larger worlds do not; a new dt reaches the chain through device memory like any other step; and a sequence of
    fused calls equals one long fused call."""
    _, part, m = bench_universe(250)
    sim = nb.SimPipeline(250, m)
    sim.set_data(part)
    sim.update(1, 0.01)
    assert sim.fused_steps() == 0                       # one step: nothing to fuse
    sim.set_data(part)
    for n_steps, dt in ((3, 0.01), (2, 0.005), (100, 0.01), (5, 0.0025)):
        sim.update(n_steps, dt)
        assert sim.fused_steps() == n_steps
    got = sim.get_data()
    assert sim.graph_stats()["cached"] == 0 and sim.graph_stats()["dt_uploads"] == 4   # no hipGraph is ever built for it
    sim.close()
    ref = nb.SimPipeline(250, m)
    ref.configure(fused_chain=0, graph=0, **matched_shape(250))
    ref.set_data(part)
    for n_steps, dt in ((3, 0.01), (2, 0.005), (100, 0.01), (5, 0.0025)):
        ref.update(n_steps, dt)
        assert ref.fused_steps() == 0
    want = ref.get_data()
    ref.close()
    assert got.tobytes() == want.tobytes()
    # explicit shape knobs ask for the per-step kernel; larger worlds stay on it
    assert run(part, m, 4, 0.01, k=1, w=16).tobytes() == run(part, m, 4, 0.01, k=1, w=16, fused_chain=0).tobytes()
    _, part333, m333 = bench_universe(333)
    big = nb.SimPipeline(333, m333)
    big.set_data(part333)
    big.update(10, 0.01)
    assert big.fused_steps() == 0
    big.close()


def test_fused_chain_against_the_reference_path():
    """The fused chain is the default for the reference harness' smallest world: one step from the bench's N = 250
    universe against float64 (through two fused steps with dt = 0: the second step's acc is the force at the unmoved
    state), and ten steps against the reference's AVX stepper on the displacement metric."""
    _, part, m = bench_universe(250)
    still = run(part, m, 2, 0.0)                        # dt = 0: nothing moves, acc = the force field, twice
    acc64, mag = ob.acc_f64(part, m)
    assert np.all(np.abs(still[:, 4:6].astype(np.float64) - acc64) <= acc_bound(acc64, mag))
    assert np.array_equal(still[:, 0:4], part[:, 0:4]) and np.array_equal(still[:, 6:8], part[:, 6:8])
    want = ob.step(part, m, 0.01, 10, kind="avx")
    sim = nb.SimPipeline(250, m)
    sim.set_data(part)
    sim.update(10, 0.01)
    assert sim.fused_steps() == 10
    got = sim.get_data()
    sim.close()
    assert rel_l2_pos(got, want) <= 1e-6 and rel_displacement(got, want, part) <= DISPLACEMENT_TOL
    # the World surface takes the same path (nbody-bench's 100-step call)
    ic = nb.make_galaxies(250, 2, seed=11037)
    w = nb.World(ic)
    w.update_gpu(0.01, 10)
    assert w.particles().tobytes() == got.tobytes()
    w.close()


@pytest.mark.parametrize("n,frac", [(1, 1.0), (2, 0.5), (64, 1.0), (65, 0.3), (129, 0.02), (130, 1.0), (257, 0.5), (300, 0.0),
                                    (511, 1.0), (512, 0.6)])
def test_fused_chain_ragged_worlds(n, frac):
    part, m = synth(n, frac, seed=7 * n)
    got = run(part, m, 3, 0.02, fused_chain=1)
    want = run(part, m, 3, 0.02, fused_chain=0, graph=0, **matched_shape(n))
    assert got.tobytes() == want.tobytes()
    if m:
        check_one_step(run(part, m, 1, 0.02, fused_chain=0, **matched_shape(n)), part, m, 0.02)


# ---------------------------------------------------------------------------------------------------------------
# lane-split launches (knob "lanes"): several source slices per receiver inside one wave, sources staged in LDS
# ---------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("lanes,w", [(2, 4), (2, 16), (4, 8), (4, 16), (8, 8), (8, 16)])
def test_lane_split_shapes_against_float64(golden, lanes, w):
    """Every lane-split instantiation against float64 and the integrator's exact roundings, at the fixtures and at
    source counts around every granule / group / slice boundary (a slice is a whole number of 8-source granules, walked in
    groups of four; trailing slices are empty), with receivers that do not fill the last workgroup."""
    for n in (4096, 333):
        part, m = ob.partition(golden(f"ic_{n}.bin"))
        sim = nb.SimPipeline(n, m)
        sim.configure(lanes=lanes, w=w)
        sim.set_data(part)
        sim.update(1, 0.01)
        shape = sim.launch_shape()
        got = sim.get_data()
        sim.close()
        assert (shape["lanes"], shape["w"], shape["k"], shape["split"]) == (lanes, w, 1, 1)
        assert shape["workgroups"] == -(-n // (64 // lanes))
        check_one_step(got, part, m, 0.01)
    for m_want in (1, 3, 4, 5, 7, 8, 9, 31, 32, 33, 63, 64, 65, 100, 255, 256, 257, 511, 513, 1000, 1031, 2049):
        n = m_want + 37
        part, m = synth(n, 1.0, seed=50 + m_want)
        part[m_want:, 6] = 0.0
        part, m = ob.partition(part)
        assert m == m_want
        got = run(part, m, 1, 0.01, lanes=lanes, w=w)
        check_one_step(got, part, m, 0.01)
    # graph replay == plain launches, and a second step reads the first one's output
    part, m = ob.partition(golden("ic_1024.bin"))
    assert run(part, m, 5, 0.01, lanes=lanes, w=w, graph=1).tobytes() == run(part, m, 5, 0.01, lanes=lanes, w=w, graph=0).tobytes()


def test_lane_split_auto_policy_and_multi_step_parity():
    """All knobs on auto: latency-bound unsharded steps (N x M <= 9e6) run lane-split -- ONE kernel per step where the
    classic model would split the sources and add a finish kernel; larger worlds, sharded pipelines, an explicit shape
    knob or the LDS-tile route keep the classic kernel.  Ten steps against the reference's AVX stepper."""
    for n, expect in ((300, True), (500, True), (2000, True), (4000, True), (10000, False), (65536, False)):
        _, part, m = bench_universe(n)
        plan = nb.plan_launch(n, m)
        assert (plan["lanes"] > 1) == expect, (n, plan)
        sim = nb.SimPipeline(n, m)
        sim.set_data(part)
        sim.update(10, 0.01)
        shape = sim.launch_shape()
        ms, launches = sim.last_step_ms()
        got = sim.get_data()
        sim.close()
        assert (shape["lanes"] > 1) == expect, (n, shape)
        if expect:
            assert (shape["lanes"], shape["w"]) == (plan["lanes"], plan["lanes_w"]) and launches == 10 and shape["split"] == 1
        if n <= 4000:
            want = ob.step(part, m, 0.01, 10, kind="avx")
            assert rel_l2_pos(got, want) <= 1e-6 and rel_displacement(got, want, part) <= DISPLACEMENT_TOL, n
    _, part, m = bench_universe(2000)
    for knobs in (dict(k=2), dict(w=8), dict(split=3), dict(unit=16), dict(variant=0), dict(lanes=1)):
        sim = nb.SimPipeline(2000, m)
        sim.configure(**knobs)
        sim.set_data(part)
        sim.update(1, 0.01)
        assert sim.launch_shape()["lanes"] == 1, knobs
        sim.close()
    g = nb.LocalShardGroup(2000, m, 2)
    g.set_data(part)
    g.step(1, 0.01)
    assert g.members[0].launch_shape()["lanes"] == 1
    g.close()
    # deterministic, and the same bits through the World surface
    a = run(part, m, 3, 0.01)
    assert a.tobytes() == run(part, m, 3, 0.01).tobytes()
    w = nb.World(nb.make_galaxies(2000, 2, seed=11037))
    w.update_gpu(0.01, 3)
    assert w.particles().tobytes() == a.tobytes()
    w.close()


@pytest.mark.parametrize("n,frac", [(1, 1.0), (2, 0.5), (15, 1.0), (16, 0.5), (17, 1.0), (63, 0.5), (64, 1.0), (65, 0.3), (130, 1.0),
                                    (257, 0.02), (300, 0.0), (1000, 0.01)])
def test_lane_split_ragged_worlds(n, frac):
    part, m = synth(n, frac, seed=3 * n + 1)
    for lanes, w in ((4, 8), (8, 16), (2, 16)):
        got = run(part, m, 1, 0.02, lanes=lanes, w=w, fused_chain=0)
        check_one_step(got, part, m, 0.02)
        two = run(part, m, 2, 0.02, lanes=lanes, w=w, fused_chain=0)
        again = run(got, m, 1, 0.02, lanes=lanes, w=w, fused_chain=0)   # step 2 from step 1's output: the same bits
        assert two.tobytes() == again.tobytes()
