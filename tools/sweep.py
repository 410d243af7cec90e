#!/usr/bin/env python3
"""Time step-kernel shapes for one libnbody_hip.so (NBODY_HIP_SO) -- tuning aid, run on the GPU box."""
import os, sys, itertools
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
shapes = [tuple(map(int, s.split(","))) for s in sys.argv[2:]] or [(1, 4, 16), (1, 2, 16), (0, 4, 16)]
ic = nb.make_galaxies(n, 2, seed=11037)
w = nb.World(ic); part = w.particles(); w.close()
m = int((part[:, 6] > 0).sum())
for (variant, k, wv) in shapes:
    sim = nb.SimPipeline(n, m); sim.configure(variant=variant, k=k, w=wv); sim.set_data(part)
    steps = 4 if n > 300000 else 30
    sim.update(2, 0.01)
    best, per_launch = 1e9, 0.0
    for _ in range(2):
        sim.update(steps, 0.01)
        ms, launches = sim.last_step_ms()
        if ms / steps < best:
            best, per_launch = ms / steps, ms / launches   # a step is `launches / steps` source passes (launches)
    print(f"{os.environ.get('NBODY_HIP_SO','default'):40s} N={n} variant={variant} k={k} w={wv}: {per_launch:9.3f} ms/launch "
          f"{best:9.3f} ms/step {n*m/(best*1e-3):.3e} int/s", flush=True)
    sim.close()
