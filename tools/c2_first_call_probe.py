import os, sys, time
sys.path.insert(0, "/root/repo")
import nbody_amd as nb
n = 65536
ic = nb.make_galaxies(n, 2, seed=11037)
w = nb.World(ic); part = w.particles(); w.close()
m = int((part[:, 6] > 0).sum())
for graph in (0, 1, 0, 1):
    sim = nb.SimPipeline(n, m); sim.configure(graph=graph); sim.set_data(part)
    sim.update(10, 0.01)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); sim.update(100, 0.01); ts.append((time.perf_counter() - t0) / 100 * 1e3)
    sim.close()
    print(f"graph={graph}: ms/step of six consecutive 100-step calls after 10 warm-up steps: " + " ".join(f"{t:.4f}" for t in ts), flush=True)
