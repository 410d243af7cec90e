#!/usr/bin/env python3
"""Host<->device hand-over cost: SetSimulationData / GetSimulationData at several N."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb
for n in (6000, 65536, 1 << 20, 1 << 22):
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic); part = w.particles(); w.close()
    m = int((part[:, 6] > 0).sum())
    sim = nb.SimPipeline(n, m); sim.set_data(part); sim.update(1, 0.01); sim.get_data()
    t0 = time.perf_counter()
    for _ in range(5): sim.set_data(part)
    t1 = time.perf_counter()
    for _ in range(5): out = sim.get_data()
    t2 = time.perf_counter()
    sim.close()
    mb = n * 32 / 1e6
    print(f"N={n:8d} ({mb:7.1f} MB): Set {1e3*(t1-t0)/5:8.3f} ms ({mb/((t1-t0)/5)/1e3:5.1f} GB/s)  Get {1e3*(t2-t1)/5:8.3f} ms ({mb/((t2-t1)/5)/1e3:5.1f} GB/s)", flush=True)
