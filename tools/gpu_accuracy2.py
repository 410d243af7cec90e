#!/usr/bin/env python3
"""Worst err/bound over launch shapes at N = 2^20 (2000-receiver sample), single pipeline and 8 local shards."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb
import oracle_binding as ob
n = 1 << 20
ic = nb.make_galaxies(n, 2, seed=11037)
w = nb.World(ic); part = w.particles(); w.close()
m = int((part[:, 6] > 0).sum())
idx = np.unique(np.random.default_rng(7).integers(0, n, 4000)).astype(np.uint32)
acc64, mag = ob.acc_f64_subset(part, m, idx)
b1 = 1e-4 * np.abs(acc64) + 1e-6 * mag
def report(tag, got):
    e = np.abs(got[idx, 4:6].astype(np.float64) - acc64)
    print(f"{tag:44s} max err/bound {np.max(e/b1):6.3f}  p99.9 {np.quantile(e/b1, 0.999):6.3f}  rms(err/mag) {np.sqrt(np.mean((e/mag)**2)):.2e}  max(err/mag) {np.max(e/mag):.2e}", flush=True)
for knobs in (dict(k=2, w=16, split=1), dict(k=2, w=4, split=11), dict(k=2, w=8, split=5), dict(k=2, w=16, split=16), dict(k=1, w=16, split=1),
              dict(k=2, w=16, split=1, variant=0), dict(k=2, w=4, split=1), dict(k=2, w=1, split=1)):
    sim = nb.SimPipeline(n, m); sim.configure(**knobs); sim.set_data(part); sim.update(1, 0.01)
    report(str(knobs), sim.get_data()); sim.close()
for overlap in (0, 1):
    g = nb.LocalShardGroup(n, m, 8, overlap=overlap); g.set_data(part); g.step(1, 0.01)
    report(f"8 local shards overlap={overlap}", g.get_data(0)); g.close()
