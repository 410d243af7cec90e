#!/usr/bin/env python3
"""us per step of N-step calls under the three graph policies (0 plain launches, 1 cached chains of <= 64 steps, 2 auto =
prebuilt canonical 32-step chain on small worlds) -- is there a gap between consecutive hipGraph launches?  Run on the GPU box.
usage: graph_policy_probe.py N..."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb
for n in [int(x) for x in sys.argv[1:]] or [500, 2000]:
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic); part = w.particles(); w.close()
    m = int((part[:, 6] > 0).sum())
    for steps in (100, 256, 1024):
        row = []
        for graph in (0, 1, 2):
            sim = nb.SimPipeline(n, m); sim.configure(graph=graph, fused_chain=0); sim.set_data(part)
            sim.update(steps, 0.01); sim.update(steps, 0.01)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); sim.update(steps, 0.01); best = min(best, (time.perf_counter() - t0) / steps)
            row.append(best * 1e6); sim.close()
        print(f"N={n:5d} {steps:5d}-step calls: plain launches {row[0]:6.2f} us/step | cached chains (<= 64 steps) {row[1]:6.2f} | auto (canonical 32-step chain) {row[2]:6.2f}", flush=True)
