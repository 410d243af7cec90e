#!/usr/bin/env python3
"""One GPU, all P shards in-process (local transport): does splitting a step into P shard launches cost anything?

Total work is identical to the single pipeline, so (time of P shard steps) / (time of one full step) - 1 is the
overhead of the sharded kernels' shapes (N/P receivers each, padded gathered sources) plus the local copies.
"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ic = nb.make_galaxies(n, 2, seed=11037)
w = nb.World(ic); part = w.particles(); w.close()
m = int((part[:, 6] > 0).sum())
sim = nb.SimPipeline(n, m); sim.set_data(part); sim.update(2, 0.01)
t0 = time.perf_counter(); sim.update(4, 0.01); t1 = time.perf_counter(); sim.close()
base = (t1 - t0) / 4
print(f"N={n} M={m}: single pipeline {base*1e3:.2f} ms/step", flush=True)
import itertools
extra = [dict(split=int(x)) for x in sys.argv[2:]]  # optional forced splits for the non-overlapped step
for P in (2, 4, 8):
    for overlap, kn in [(0, {}), (1, {})] + [(0, e) for e in extra]:
        g = nb.LocalShardGroup(n, m, P, overlap=overlap, **kn)
        g.set_data(part); g.step(1, 0.01)
        t0 = time.perf_counter(); g.step(3, 0.01); t1 = time.perf_counter()
        per = (t1 - t0) / 3
        shape = g.members[0].launch_shape()
        g.close()
        print(f"  P={P} overlap={overlap} {kn}: {per*1e3:.2f} ms per step of all shards ({per/base:.3f}x)  => ideal per-GPU step {per/P*1e3:.2f} ms, shape {shape}", flush=True)
