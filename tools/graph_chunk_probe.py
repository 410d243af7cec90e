#!/usr/bin/env python3
"""How short can a replayed hipGraph chunk be before the per-launch cost shows?  96 steps enqueued without a host wait
in between, as plain launches, as one 96-step request (64 + 32 chunks) and as 6 x 16 / 12 x 8 / 3 x 32 replays of a
cached short chain; plus what the FIRST use of each chunk length costs (build + instantiate on the calling thread)."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb
for n in [int(x) for x in sys.argv[1:]] or [250, 1000, 4000, 10000]:
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic); part = w.particles(); w.close()
    m = int((part[:, 6] > 0).sum())
    row = []
    for label, graph, chunk in (("plain", 0, 96), ("graph 64+32", 1, 96), ("graph 3x32", 1, 32), ("graph 6x16", 1, 16), ("graph 12x8", 1, 8)):
        sim = nb.SimPipeline(n, m); sim.configure(graph=graph, timing=0); sim.set_data(part)
        sim.update(2, 0.01)
        t0 = time.perf_counter(); sim.step_async(chunk, 0.01); first_enqueue = (time.perf_counter() - t0) * 1e6; sim.sync()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(96 // chunk): sim.step_async(chunk, 0.01)
            sim.sync()
            best = min(best, (time.perf_counter() - t0) * 1e6 / 96)
        sim.close()
        row.append(f"{label}: {best:5.2f} us/step (first enqueue {first_enqueue:6.0f} us)")
    print(f"N={n:6d}: " + " | ".join(row), flush=True)
