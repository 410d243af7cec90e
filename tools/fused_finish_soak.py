#!/usr/bin/env python3
"""Soak of the fused finish: long runs and random shapes, fused vs the two-kernel form, bit for bit."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb
import gpu_common as T

bad = 0
def run(part, m, steps, **knobs):
    sim = nb.SimPipeline(part.shape[0], m); sim.configure(**knobs); sim.set_data(part)
    left = steps
    while left > 0:
        c = min(left, 1000); sim.update(c, 0.001); left -= c
    out = sim.get_data(); sim.close(); return out

# 1. long runs at the sizes the auto rule would cover: every step's tile finishes must see all parts
for n, steps in ((10000, 20000), (20000, 6000), (50000, 1500), (1 << 20, 12)):
    ic = nb.make_galaxies(n, 2, seed=11037); w = nb.World(ic); part = w.particles(); w.close(); m = int((part[:, 6] > 0).sum())
    for graph in (0, 1):
        a = run(part, m, steps, graph=graph, fused_finish=0); b = run(part, m, steps, graph=graph, fused_finish=1)
        same = a.tobytes() == b.tobytes(); bad += not same
        print(f"long run N={n} steps={steps} graph={graph}: {'identical' if same else 'DIFFER'}", flush=True)
# 2. random worlds x shapes x passes
rng = np.random.default_rng(4242)
for case in range(300):
    n = int(rng.choice([130, 777, 1500, 3000, 4097, 6000, 9000, 12000, 20011]))
    part, m = T.synth(n, float(rng.choice([0.05, 0.3, 0.5, 1.0])), seed=int(rng.integers(1 << 30)), extent=float(rng.choice([1e3, 1e5])))
    knobs = dict(k=int(rng.choice([1, 2])), w=int(rng.choice([4, 8, 16])), split=int(rng.integers(2, 17)), unit=int(rng.choice([0, 8, 16, 32, 64])),
                 passes=int(rng.choice([0, 1, 2, 3])), graph=int(rng.choice([0, 1])), lanes=1)
    steps = int(rng.choice([1, 2, 5, 17]))
    a = T.run(part, m, steps, 0.01, fused_finish=0, **knobs); b = T.run(part, m, steps, 0.01, fused_finish=1, **knobs)
    if a.tobytes() != b.tobytes():
        bad += 1; print("DIFFER", n, m, knobs, steps, flush=True)
print("fused finish soak: failures", bad)
