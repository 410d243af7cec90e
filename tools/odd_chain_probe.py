#!/usr/bin/env python3
"""A frame loop that asks for the same ODD chain length flips the ping-pong phase on every call.  With the phase in
the cache key (pipeline.hip find_graph) the pipeline holds one instantiated chain per phase and replays them untouched;
a loop that also changes dt on every call costs one 4-byte in-stream write per call (the step size lives in device
memory; round 2's first version re-patched every node instead: +5 us per call for 3 steps, +12 us for 7)."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb
n_calls = 200
for n, chain in ((4000, 3), (4000, 7), (1000, 3), (20000, 3)):
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic); part = w.particles(); w.close()
    m = int((part[:, 6] > 0).sum())
    row = {}
    for label, graph, dts in (("graph, same dt", 1, (0.01, 0.01)), ("graph, dt changes every call", 1, (0.01, 0.005, 0.0025)), ("plain launches", 0, (0.01, 0.01))):
        sim = nb.SimPipeline(n, m); sim.configure(graph=graph); sim.set_data(part)
        for i in range(6): sim.update(chain, dts[i % len(dts)])
        t0 = time.perf_counter()
        for i in range(n_calls): sim.update(chain, dts[i % len(dts)])
        row[label] = ((time.perf_counter() - t0) / n_calls * 1e6, sim.graph_stats())
        sim.close()
    print(f"N={n:6d} chain of {chain}: " + " | ".join(f"{k}: {v[0]:7.1f} us/call cached={v[1]['cached']} dt_uploads={v[1]['dt_uploads']}" for k, v in row.items()), flush=True)
