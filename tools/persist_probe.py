#!/usr/bin/env python3
"""The persistent-launch experiment: mid-N steps with FEWER, LONGER-LIVED waves -- a launch of 1/P as
many workgroups as (receiver tile, source part) work items, every workgroup walking P items (tuning hook "persist" = P;
step_kernel<..., PERSIST = true>), so that dispatch ramp and end-of-kernel write-back are paid by fewer workgroups.

For each N: the auto shape (P = 1: the classic launch) against P = 2, 3, 4, 6, 8 on the same (k, w, split, unit), and against
persistent launches over FINER items (split raised so that the item count stays near the chip's capacity x P).
(1) bits: 40 steps persistent vs classic, plain launches and hipGraph, must be identical (an item runs the code a classic
workgroup runs); (2) microseconds per step, fastest of 5 calls of 100 steps: plain launches (graph = 0) and cached graph
replays (graph = 1); (3) the share of the floor nbody-bench prints (N*M at the large-N rate + 1.7 us per dependent kernel)."""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb   # noqa: E402

assert nb.hip_lib().nb_hip_tuning_build(), "needs a `make -C nbody_amd/csrc TUNING=1` build (the persistent kernels are not in the shipped library)"

RATE = float(os.environ.get("FLOOR_RATE", "5.54e12"))


def universe(n):
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic)
    part = w.particles()
    w.close()
    return part, int((part[:, 6] > 0).sum())


def run(part, m, steps, **knobs):
    sim = nb.SimPipeline(part.shape[0], m)
    sim.configure(**knobs)
    sim.set_data(part)
    sim.update(steps, 0.01)
    out = sim.get_data()
    shape = sim.launch_shape()
    sim.close()
    return out, shape


def us_per_step(part, m, **knobs):
    sim = nb.SimPipeline(part.shape[0], m)
    sim.configure(**knobs)
    sim.set_data(part)
    sim.update(100, 0.01)
    sim.update(100, 0.01)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        sim.update(100, 0.01)
        best = min(best, (time.perf_counter() - t0) / 100 * 1e6)
    sim.close()
    return best


for n in [int(x) for x in sys.argv[1:]] or (10000, 20000, 50000):
    part, m = universe(n)
    base, shape = run(part, m, 40, graph=0)
    fixed = dict(k=shape["k"], w=shape["w"], split=shape["split"], unit=shape["unit"])
    floor = n * m / RATE * 1e6 + 1.7
    t0 = {g: us_per_step(part, m, graph=g) for g in (0, 1)}
    print(f"N={n} M={m} auto shape {shape}: classic launch {t0[0]:.2f} us/step plain, {t0[1]:.2f} graph replays; floor {floor:.2f} us "
          f"({floor / t0[1]:.1%})", flush=True)
    for persist in (2, 3, 4, 6, 8):
        got, sh = run(part, m, 40, graph=0, persist=persist, **fixed)
        same = got.tobytes() == base.tobytes() and run(part, m, 40, graph=1, persist=persist, **fixed)[0].tobytes() == base.tobytes()
        t = {g: us_per_step(part, m, graph=g, persist=persist, **fixed) for g in (0, 1)}
        print(f"   persist={persist} workgroups={sh['workgroups']:5d} bits {'identical' if same else 'DIFFER'} | plain {t[0]:7.2f} "
              f"({t[0] - t0[0]:+.2f}) | graph replays {t[1]:7.2f} ({t[1] - t0[1]:+.2f}) | {floor / t[1]:.1%} of floor", flush=True)
    # finer items: more source parts per tile, the same number of workgroups as the classic launch or half of it
    for split in sorted({min(16, shape["split"] * 2), 16} - {shape["split"]}):
        for persist in (2, 4):
            knobs = dict(fixed, split=split, persist=persist)
            ref, _ = run(part, m, 40, graph=0, **dict(fixed, split=split))
            got, sh = run(part, m, 40, graph=0, **knobs)
            t = {g: us_per_step(part, m, graph=g, **knobs) for g in (0, 1)}
            print(f"   split={split:2d} persist={persist} workgroups={sh['workgroups']:5d} bits {'identical' if got.tobytes() == ref.tobytes() else 'DIFFER'} "
                  f"(vs the classic launch of that split) | plain {t[0]:7.2f} ({t[0] - t0[0]:+.2f}) | graph replays {t[1]:7.2f} "
                  f"({t[1] - t0[1]:+.2f}) | {floor / t[1]:.1%} of floor", flush=True)
