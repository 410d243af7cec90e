#!/usr/bin/env python3
"""A/B of explicit launch shapes against the auto shape at one N (run on the GPU box): ms per step of cached-graph replays.
usage: shape_ab.py N "k,w,split[,unit]" ..."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb
n = int(sys.argv[1])
ic = nb.make_galaxies(n, 2, seed=11037)
w = nb.World(ic); part = w.particles(); w.close()
m = int((part[:, 6] > 0).sum())
steps = 100 if n <= 70000 else 20
for spec in ["auto"] + sys.argv[2:] + ["auto"]:
    knobs = {} if spec == "auto" else dict(zip(("k", "w", "split", "unit"), map(int, spec.split(","))))
    sim = nb.SimPipeline(n, m); sim.configure(graph=1, **knobs); sim.set_data(part)
    sim.update(steps, 0.01); sim.update(steps, 0.01)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); sim.update(steps, 0.01); best = min(best, (time.perf_counter() - t0) / steps)
    sh = sim.launch_shape(); sim.close()
    print(f"N={n} M={m} {spec:12s}: {best*1e3:9.4f} ms/step {n*m/best:.4e} int/s  frac {n*m/best*14/157.3e12:.4f}  shape {sh}", flush=True)
