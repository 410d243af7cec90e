#!/usr/bin/env python3
"""Lane-split shapes (knob "lanes") against the auto launch shape over the reference harness' sizes: accuracy of one step
vs float64 (the oracle is the checker here, as in the tests) and microseconds per step of cached-graph replays.
usage: lane_probe.py N..."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb, oracle_binding as ob


def timed(n, m, part, **knobs):
    sim = nb.SimPipeline(n, m); sim.configure(fused_chain=0, **knobs); sim.set_data(part)
    steps = 200 if n <= 20000 else 40
    sim.update(steps, 0.01); sim.update(steps, 0.01)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); sim.update(steps, 0.01); best = min(best, (time.perf_counter() - t0) / steps)
    sh = sim.launch_shape(); sim.close()
    return best * 1e6, sh


for n in [int(x) for x in sys.argv[1:]] or [500, 800, 1200, 2000, 4000, 10000]:
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic); part = w.particles(); w.close()
    m = int((part[:, 6] > 0).sum())
    acc64, mag = ob.acc_f64(part, m)
    bound = 1e-4 * np.abs(acc64) + 1e-6 * mag
    auto_us, auto_sh = timed(n, m, part, lanes=1)
    rows = []
    for lanes in (2, 4, 8):
        for wv in ((4, 8, 16) if lanes < 8 else (8, 16)):
            sim = nb.SimPipeline(n, m); sim.configure(lanes=lanes, w=wv); sim.set_data(part); sim.update(1, 0.01)
            got = sim.get_data(); sh = sim.launch_shape(); sim.close()
            if sh["lanes"] != lanes:
                continue
            worst = float(np.max(np.abs(got[:, 4:6].astype(np.float64) - acc64) / bound))
            v = part[:, 2:4] + got[:, 4:6] * np.float32(0.01)
            exact = bool(np.array_equal(got[:, 2:4], v) and np.array_equal(got[:, 0:2], part[:, 0:2] + v * np.float32(0.01)))
            us, _ = timed(n, m, part, lanes=lanes, w=wv)
            rows.append((us, lanes, wv, worst, exact, sh["workgroups"]))
    rows.sort()
    print(f"N={n:6d} M={m:5d}: auto {auto_us:7.2f} us (k={auto_sh['k']} w={auto_sh['w']} split={auto_sh['split']} unit={auto_sh['unit']}) | " +
          " | ".join(f"lanes={l} w={wv}: {us:6.2f} us err/bound {worst:.2f}{'' if exact else ' INTEGRATOR MISMATCH'} wg={wg}" for us, l, wv, worst, exact, wg in rows), flush=True)
