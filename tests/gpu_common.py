"""Helpers shared by the GPU test files (TEST INFRASTRUCTURE): the stated tolerances as code, one-step and multi-step
checkers against the oracle, synthetic worlds, the bench universe, and the subprocess wrappers of the harness tests."""
import ctypes as C  # noqa: F401
import os
import subprocess
import sys
import time  # noqa: F401

import numpy as np
import pytest  # noqa: F401

import nbody_amd as nb
import oracle_binding as ob


SHAPES = [(0, 0), (1, 1), (1, 16), (2, 4), (2, 1), (1, 4), (2, 16), (2, 8)]   # K = 4 / W = 2 exist in TUNING=1 builds only


def acc_bound(acc64, mag):
    return 1e-4 * np.abs(acc64) + 1e-6 * mag


def check_one_step(got, part, m, dt, want=None):
    """got = device state one step after `part`; checks acc against float64 and vel/pos against the AVX path."""
    acc64, mag = ob.acc_f64(part, m)
    bound = acc_bound(acc64, mag)
    err = np.abs(got[:, 4:6].astype(np.float64) - acc64)
    assert np.all(err <= bound), f"acc outside tolerance: worst ratio {np.max(err / bound):.3f}"
    # the integrator is exact fp32 arithmetic on the device's own acc, with the reference's roundings
    # (vel += acc*dt; pos += vel*dt; mul then add, reference sim_cpu.c:191-193): bit-exact
    v = part[:, 2:4] + got[:, 4:6] * np.float32(dt)
    p = part[:, 0:2] + v * np.float32(dt)
    assert np.array_equal(got[:, 2:4], v), "velocity is not vel + acc*dt in fp32"
    assert np.array_equal(got[:, 0:2], p), "position is not pos + vel*dt in fp32"
    # against the reference AVX path: both sit within the same bound of the float64 sum
    if want is None:
        want = ob.step(part, m, dt, 1)
    e_ref = np.abs(want[:, 4:6].astype(np.float64) - acc64)
    assert np.all(np.abs(got[:, 4:6].astype(np.float64) - want[:, 4:6]) <= bound + e_ref)
    assert np.array_equal(got[:, 6:8], part[:, 6:8]), "mass / radius must pass through untouched"


def run(part, m, n, dt, **knobs):
    sim = nb.SimPipeline(part.shape[0], m)
    sim.configure(**knobs)
    sim.set_data(part)
    sim.update(n, dt)
    out = sim.get_data()
    sim.close()
    return out


def rel_displacement(got, want, start):
    """Multi-step parity metric relative to what the steps MOVED, not to where the particles are:
    |(pos - pos0)_gpu - (pos - pos0)_ref| / |(pos - pos0)_ref| over all particles.  Relative to the positions
    themselves (1e4..1e6) ten steps at dt = 0.01 are a 1e-4 perturbation, so "rel L2 of pos <= 1e-6" would still pass
    with gravity switched off (5.5e-4 on this metric's scale); the reference's own sequential and AVX summation orders
    differ by 8.7e-7 here (reference world.c:99-110 semantics, ten calls of the step)."""
    p0 = start[:, 0:2].astype(np.float64)
    dg = got[:, 0:2].astype(np.float64) - p0
    dw = want[:, 0:2].astype(np.float64) - p0
    return float(np.linalg.norm(dg - dw) / np.linalg.norm(dw))


DISPLACEMENT_TOL = 1e-4   # stated multi-step tolerance (README / DESIGN.md section 5); observed ~1e-6


def synth(n, frac_massive=0.5, seed=0, extent=1.0e4):
    rng = np.random.default_rng(seed)
    a = np.zeros((n, 8), dtype=np.float32)
    a[:, 0:2] = rng.standard_normal((n, 2)) * extent
    a[:, 2:4] = rng.standard_normal((n, 2)) * 10
    massive = rng.random(n) < frac_massive
    a[:, 7] = np.where(massive, 1.5 + 8 * rng.random(n), 0.5)
    a[:, 6] = np.where(massive, 41.9 * a[:, 7] ** 3, 0.0)
    return ob.partition(a)

def avx_steps(part, m, schedule):
    """The reference AVX stepper (bit-exact restatement, world.c:99-110 semantics) over a schedule of (steps, dt) calls."""
    state = part
    for n, dt in schedule:
        state = ob.step(state, m, dt, n)
    return state


def f64_steps(part, m, schedule):
    """The float64 stepper (terms, sums, state and integrator in double: oracle orc_step_f64) over the same schedule."""
    state = part
    for n, dt in schedule:
        state = ob.step(state, m, dt, n, kind="f64")
    return state


SEGMENT_STEPS = 10   # the stated multi-step tolerance is defined over at most ten steps from an identical state


def assert_anchored(got, part, m, schedule, label="", **knobs):
    """Multi-step anchor of a GPU trajectory of ANY length against the reference CPU path (world.c:99-110,
    sim_cpu.c:156-194), with float64 as the tie-breaker ("closer to fp64 than the AVX path is, is
    acceptable") -- no looser bound for long chains and no part of a schedule left out.

    Up to ten steps: `got` itself against the AVX stepper at the stated tolerance (1e-4 of what the steps moved), and no
    further from the float64 trajectory than 1.5 x the AVX stepper's own distance from it.

    Longer schedules: the N-body system is chaotic (a close encounter in the 333-particle fixture multiplies any
    difference by ~4 000 between steps 130 and 343: the reference's own AVX build ends 1.8e-3 from the float64 trajectory
    there, this engine 2.9e-3), so end-to-end distances of two fp32 implementations are O(1) multiples of each other by
    chance and cannot carry a bound.  Instead the WHOLE schedule is re-walked on a second pipeline in calls of at most ten
    steps (plain launches; `knobs` select the launch shape the trajectory under test is bit-equal to), and every segment
    is anchored FROM THE GPU'S OWN STATE at its start: GPU segment vs AVX segment at 1e-4, GPU-f64 <= 1.5 x AVX-f64.  The
    re-walk must end on `got` bit for bit, which ties the checked segments to the trajectory under test.  The end-to-end
    distances are printed for the record (pytest -rP)."""
    total = sum(n for n, _ in schedule)

    def check_segment(end_state, start_state, n, dt, what):
        avx = ob.step(start_state, m, dt, n)
        f64 = ob.step(start_state, m, dt, n, kind="f64")
        d_pair = rel_displacement(end_state, avx, start_state)
        d_gpu, d_avx = rel_displacement(end_state, f64, start_state), rel_displacement(avx, f64, start_state)
        assert d_pair <= DISPLACEMENT_TOL, (label, what, d_pair)
        assert d_gpu <= 1.5 * d_avx + 1e-9, (label, what, d_gpu, d_avx)
        assert np.array_equal(end_state[:, 6:8], avx[:, 6:8])
        return d_pair, d_gpu, d_avx

    if total <= SEGMENT_STEPS and len(schedule) == 1:
        d = check_segment(got, part, schedule[0][0], schedule[0][1], f"{total} steps")
        print(f"[anchor] {label} {total} steps: gpu-avx {d[0]:.3e}  gpu-f64 {d[1]:.3e}  avx-f64 {d[2]:.3e}")
        return
    walker = nb.SimPipeline(part.shape[0], m)
    walker.configure(**dict(dict(graph=0), **knobs))
    walker.set_data(part)
    state, done, worst = part, 0, (0.0, 0.0, 0.0)
    for n, dt in schedule:
        left = n
        while left > 0:
            k = min(left, SEGMENT_STEPS)
            walker.update(k, dt)
            nxt = walker.get_data()
            d = check_segment(nxt, state, k, dt, f"steps {done}..{done + k} of {total}")
            worst = tuple(max(a, b) for a, b in zip(worst, d))
            state, done, left = nxt, done + k, left - k
    walker.close()
    assert state.tobytes() == got.tobytes(), (label, "the re-walked schedule does not end on the trajectory under test")
    f64, avx = f64_steps(part, m, schedule), avx_steps(part, m, schedule)
    print(f"[anchor] {label} {total} steps in segments of <= {SEGMENT_STEPS}: worst segment gpu-avx {worst[0]:.3e}  gpu-f64 {worst[1]:.3e}  "
          f"avx-f64 {worst[2]:.3e}; end to end (chaotic, not asserted): gpu-f64 {rel_displacement(got, f64, part):.3e}  "
          f"avx-f64 {rel_displacement(avx, f64, part):.3e}  gpu-avx {rel_displacement(got, avx, part):.3e}")

def matched_shape(n):
    """The per-step launch shape whose summation order the one-workgroup chain reproduces."""
    tiles = 1 if n <= 128 else 2 if n <= 256 else 4
    return dict(k=2, w=16 // tiles, split=1, unit=8)


def bench_universe(n):
    """srand(11037) MakeGalaxies(n, 2) through CreateWorld's partition: the bench's universe at size n."""
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic)
    part = w.particles()
    w.close()
    return ic, part, int((part[:, 6] > 0).sum())


def rel_l2_pos(got, want):
    d = got[:, 0:2].astype(np.float64) - want[:, 0:2].astype(np.float64)
    return float(np.linalg.norm(d) / np.linalg.norm(want[:, 0:2].astype(np.float64)))


# |acc_gpu - acc_avx| <= C(N) * sum_j |contribution_j| on a 2 000-receiver sample of one step, and K steps against the AVX
# stepper relative to what they moved.  Constants = ~3x what tools/gpu_vs_avx.py measured (profiles/r04_gpu_vs_avx.txt);
# the deviation is the AVX path's own sequential-sum error, which grows with M (the float64 columns there show it).
GPU_VS_AVX = {65536: (1.0e-4, 10), 262144: (5.0e-4, 10), 1 << 20: (1.0e-3, 2)}


def _bench_ranks(args, env=None, timeout=600):
    exe = os.path.join(nb.LIB_DIR, "nbody-bench")
    e = dict(os.environ, OMP_NUM_THREADS="2")
    e.update(env or {})
    return subprocess.run([exe] + args, env=e, capture_output=True, text=True, timeout=timeout)

_MULTI_PROC_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
root = sys.argv[1]; out_path = sys.argv[2]; n = int(sys.argv[3]); overlap = int(sys.argv[4])
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
import nbody_amd as nb, oracle_binding as ob
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ic = np.fromfile(os.path.join(root, "tests", "golden", f"ic_{n}.bin"), dtype=np.float32).reshape(-1, 8)
part, m = ob.partition(ic)
calls = []
def gather(rows, r, nr):
    calls.append(rows.shape)
    mine = torch.from_numpy(rows[r].copy())
    parts = [torch.empty_like(mine) for _ in range(nr)]
    dist.all_gather(parts, mine)
    for q in range(nr):
        if q != r: rows[q] = parts[q].numpy()
res = {}
for tag, knobs in (("w1", dict(w=1, k=1)), ("auto", dict())):
    sim = nb.SimPipeline(n, m, rank=rank, nranks=world, allgather=gather)
    sim.configure(overlap=overlap, **knobs)
    sim.set_data(part)
    sim.update(2, 0.01); sim.update(1, 0.01)
    steps, k_ms, c_ms = sim.step_breakdown()
    assert steps == 1 and k_ms > 0 and c_ms > 0
    info = sim.comm_info()
    assert not info["owns_comm"] and info["nranks"] == world and info["rank"] == rank and "host" in info["rccl_lib"]
    res[tag] = sim.get_data()          # collective: every rank gets the full array
    sim.close()
plan = nb.shard_plan(n, m, rank, world)
assert len(calls) == 2 * (3 + 1) and calls[0] == (world, plan["mass_chunk"] * 8)
# every rank must hold the same bytes
for tag in res:
    mine = torch.from_numpy(res[tag].view(np.uint8).reshape(-1).copy())
    ref = mine.clone(); dist.broadcast(ref, src=0)
    assert bool((mine == ref).all()), f"rank {rank} differs from rank 0 ({tag})"
if rank == world - 1:                  # written by the LAST rank: a rank > 0 produced the checked bytes
    np.save(out_path, np.stack([res["w1"], res["auto"]]))
dist.barrier(); dist.destroy_process_group()
'''


__all__ = [n for n in dir() if not n.startswith("__")]
