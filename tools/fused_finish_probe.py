#!/usr/bin/env python3
"""The fused-finish experiment: split steps whose LAST-ARRIVING workgroup per receiver tile adds the
parts and integrates inside the step kernel (agent-scope sc1 stores / loads of the parts + one ticket per tile) instead
of a second, dependent finish kernel.  Knob "fused_finish" (default 0).

For each N: (1) bits -- 60 steps fused vs the two-kernel form, plain launches and hipGraph, must be identical;
(2) microseconds per step, fastest of 5 calls of 100 steps: plain launches (graph = 0) and cached graph replays (graph = 1),
two-kernel vs fused; (3) the share of the floor nbody-bench prints (N*M at the large-N rate + 1.7 us per dependent kernel)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb   # noqa: E402

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


for n in [int(x) for x in sys.argv[1:]] or (6000, 10000, 14000, 20000, 30000, 50000, 100000):
    part, m = universe(n)
    a, shape = run(part, m, 60, graph=0, fused_finish=0)
    same = all(run(part, m, 60, graph=g, fused_finish=1)[0].tobytes() == a.tobytes() for g in (0, 1))
    if shape["split"] <= 1:
        print(f"N={n:7d} M={m:6d} shape {shape}: unsplit, nothing to fuse")
        continue
    t = {(g, f): us_per_step(part, m, graph=g, fused_finish=f) for g in (0, 1) for f in (0, 1)}
    floor2 = n * m / RATE * 1e6 + 2 * 1.7
    print(f"N={n:7d} M={m:6d} k={shape['k']} w={shape['w']} split={shape['split']:2d} bits {'identical' if same else 'DIFFER'} | "
          f"plain launches: two kernels {t[(0, 0)]:7.2f} fused {t[(0, 1)]:7.2f} ({t[(0, 1)] - t[(0, 0)]:+.2f}) | "
          f"graph replays: two kernels {t[(1, 0)]:7.2f} fused {t[(1, 1)]:7.2f} ({t[(1, 1)] - t[(1, 0)]:+.2f}) | "
          f"floor {floor2:7.2f} us: {floor2 / t[(1, 0)]:.1%} -> {floor2 / t[(1, 1)]:.1%}", flush=True)
