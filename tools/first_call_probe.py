#!/usr/bin/env python3
"""The reference harness' call pattern on a fresh pipeline (bench.c:25-35): update(10) then ONE timed update(100); also
the second and third 100-step calls.  Default knobs (graph = 2: canonical prebuilt chain on small worlds) vs graph = 0."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb
for n in [int(x) for x in sys.argv[1:]] or [250, 1000, 2000, 4000, 10000, 20000]:
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic); part = w.particles(); w.close()
    m = int((part[:, 6] > 0).sum())
    row = []
    for graph in (2, 0):
        best = None
        for rep in range(3):        # three fresh pipelines: the first timed call of each
            sim = nb.SimPipeline(n, m); sim.configure(graph=graph, timing=0); sim.set_data(part)
            sim.update(10, 1.0)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); sim.update(100, 1.0); ts.append((time.perf_counter() - t0) * 1e4)
            stats = sim.graph_stats()
            sim.close()
            best = ts if best is None or ts[0] < best[0] else best
        row.append(f"graph={graph}: " + " ".join(f"{t:6.2f}" for t in best) + f" us/step (calls 1-3, cached={stats['cached']})")
    print(f"N={n:6d}: " + " | ".join(row), flush=True)
