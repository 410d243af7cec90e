#!/usr/bin/env python3
"""Exhaustive (k, w, split) scan at given N vs the auto launch shape -- tuning aid for choose_shape's cost model."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb


def time_shape(n, m, part, **knobs):
    sim = nb.SimPipeline(n, m); sim.configure(**knobs); sim.set_data(part)
    steps = 30 if n > 30000 else 200
    sim.update(3, 0.01)
    best = 1e9
    for _ in range(3):
        sim.update(steps, 0.01); ms, _l = sim.last_step_ms(); best = min(best, ms / steps)
    sh = sim.launch_shape(); sim.close()
    return best * 1e3, sh


for n in [int(x) for x in sys.argv[1:]] or [2000, 4000, 6000, 10000, 20000, 50000]:
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic); part = w.particles(); w.close()
    m = int((part[:, 6] > 0).sum())
    auto_us, auto_sh = time_shape(n, m, part)
    rows = []
    units = [int(x) for x in os.environ.get("SWEEP_UNITS", "64,16,8").split(",")]
    for k in (1, 2):
        for wv in (4, 8, 16):
            for sp in (1, 2, 3, 4, 5, 6, 8, 10, 13, 16):
                for un in units:
                    us, sh = time_shape(n, m, part, k=k, w=wv, split=sp, unit=un)
                    rows.append((us, k, wv, sp, un, sh["workgroups"]))
    rows.sort()
    if os.environ.get("SWEEP_DUMP"):
        with open(os.environ["SWEEP_DUMP"], "a") as f:
            for us, k, wv, sp, un, wg in rows:
                f.write(f"{n} {m} {k} {wv} {sp} {un} {wg} {us:.2f}\n")
    print(f"N={n} M={m}: auto {auto_us:.1f} us (k={auto_sh['k']} w={auto_sh['w']} split={auto_sh['split']} unit={auto_sh['unit']} wg={auto_sh['workgroups']}); "
          f"ideal at 5.3e12/s {n*m/5.3e12*1e6:.1f} us; best five: " +
          " | ".join(f"{us:.1f} us k={k} w={wv} split={sp} unit={un} wg={wg}" for us, k, wv, sp, un, wg in rows[:5]), flush=True)
    best64 = [r for r in rows if r[4] == 64][:2]
    print("   best with unit=64: " + " | ".join(f"{us:.1f} us k={k} w={wv} split={sp}" for us, k, wv, sp, un, wg in best64), flush=True)
