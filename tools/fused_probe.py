#!/usr/bin/env python3
"""One-workgroup chain vs per-step launches of the matching shape: bits and microseconds per step (run on the GPU box).
usage: fused_probe.py check N STEPS | time N..."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb


def universe(n):
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic); part = w.particles(); w.close()
    return part, int((part[:, 6] > 0).sum())


def matched(n):
    tiles = 1 if n <= 128 else 2 if n <= 256 else 4
    return dict(k=2, w=16 // tiles, split=1, unit=8)


def run(part, m, steps, **knobs):
    sim = nb.SimPipeline(part.shape[0], m); sim.configure(**knobs); sim.set_data(part)
    print("  update", knobs, flush=True)
    sim.update(steps, 0.01)
    out = sim.get_data(); fused = sim.fused_steps(); sim.close()
    return out, fused


mode = sys.argv[1]
if mode == "msweep":
    # fixed cost per step of the one-workgroup chain: N receivers, the first M of them massive
    n = int(sys.argv[2])
    rng = np.random.default_rng(1)
    for m in [int(x) for x in sys.argv[3:]]:
        a = np.zeros((n, 8), dtype=np.float32)
        a[:, 0:2] = rng.standard_normal((n, 2)) * 1e4
        a[:, 7] = 2.0
        a[:m, 6] = 1e3
        sim = nb.SimPipeline(n, m); sim.configure(fused_chain=1); sim.set_data(a)
        sim.update(100, 0.01)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); sim.update(2000, 0.01); best = min(best, (time.perf_counter() - t0) / 2000)
        sim.close()
        print(f"N={n} M={m:4d}: {best*1e6:6.2f} us/step", flush=True)
elif mode == "check":
    n, steps = int(sys.argv[2]), int(sys.argv[3])
    part, m = universe(n)
    print(f"N={n} M={m}: plain matched shape", flush=True)
    want, f0 = run(part, m, steps, fused_chain=0, graph=0, **matched(n))
    print("fused", flush=True)
    got, f1 = run(part, m, steps, fused_chain=1)
    print(f"fused steps {f1} (plain {f0}); bit-equal: {got.tobytes() == want.tobytes()}", flush=True)
    sys.exit(0 if got.tobytes() == want.tobytes() and f1 == steps else 1)
else:
    for n in [int(x) for x in sys.argv[2:]]:
        part, m = universe(n)
        row = []
        for knobs in (dict(fused_chain=0), dict(fused_chain=1), dict(fused_chain=0, **matched(n))):
            sim = nb.SimPipeline(n, m); sim.configure(**knobs); sim.set_data(part)
            sim.update(100, 0.01)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); sim.update(1000, 0.01); best = min(best, (time.perf_counter() - t0) / 1000)
            row.append(best * 1e6); sim.close()
        print(f"N={n:5d} M={m:5d} pairs={n*m:8d}: per-step launches (auto shape) {row[0]:6.2f} us/step | one-workgroup chain {row[1]:6.2f} | "
              f"per-step launches, matched shape {row[2]:6.2f}", flush=True)
