#!/usr/bin/env python3
"""What a hipGraph chain costs to build and what it saves per step: first and later PerformSimUpdate(100) calls,
graph on/off, wall clock (the reference's nbody-bench times ONE such call per backend)."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb
for n in [int(x) for x in sys.argv[1:]] or [1000, 4000, 10000, 20000, 100000]:
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic); part = w.particles(); w.close()
    m = int((part[:, 6] > 0).sum())
    row = []
    for graph in (1, 0):
        sim = nb.SimPipeline(n, m); sim.configure(graph=graph); sim.set_data(part)
        sim.update(10, 1.0)                      # the harness' warm-up call
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); sim.update(100, 1.0); ts.append((time.perf_counter() - t0) * 1e4)
        sim.close()
        row.append(ts)
    print(f"N={n:7d}: graph on: " + " ".join(f"{t:7.1f}" for t in row[0]) + "  us/step (calls 1-4) | graph off: " + " ".join(f"{t:7.1f}" for t in row[1]), flush=True)
