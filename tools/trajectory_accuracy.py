#!/usr/bin/env python3
"""Multi-step drift vs a float64 integration: GPU path next to the reference AVX path (same ICs, dt = 0.01)."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb
import oracle_binding as ob
for n in (4096, 16384):
    ic = nb.make_galaxies(n, 2, seed=11037)
    part, m = ob.partition(ic)
    for steps in (1, 10, 100):
        truth = ob.step(part, m, 0.01, steps, kind="f64").astype(np.float64)
        avx = ob.step(part, m, 0.01, steps, kind="avx").astype(np.float64)
        sim = nb.SimPipeline(n, m); sim.set_data(part); sim.update(steps, 0.01); gpu = sim.get_data().astype(np.float64); sim.close()
        def rel(a): return np.linalg.norm(a[:, 0:2] - truth[:, 0:2]) / np.linalg.norm(truth[:, 0:2])
        def relv(a): return np.linalg.norm(a[:, 2:4] - truth[:, 2:4]) / np.linalg.norm(truth[:, 2:4])
        print(f"N={n:6d} steps={steps:4d}: rel-L2 position error vs float64  GPU {rel(gpu):.3e}  AVX {rel(avx):.3e} | velocity  GPU {relv(gpu):.3e}  AVX {relv(avx):.3e}", flush=True)
