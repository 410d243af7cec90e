#!/usr/bin/env python3
"""Seeded fuzz of the small-world paths (lane-split launches, the one-workgroup chain, graph vs plain launches) beyond what
the test suite runs: 480 random worlds x knobs, each checked like the GPU suite does (tests/gpu_common.py) (one step against float64
with an exact integrator, graph == plain bitwise, several steps against the reference AVX stepper on the displacement
metric).  Run on the GPU box; prints the failing cases, then the count."""
import os, sys, numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb, oracle_binding as ob
import gpu_common as T
bad = 0
for block in range(100, 160):
    rng = np.random.default_rng(block)
    for case in range(8):
        n = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 777, 1024, 1500, 2111, 3000, 5000]))
        frac = float(rng.choice([0.0, 0.02, 0.3, 0.5, 1.0]))
        part, m = T.synth(n, frac, seed=int(rng.integers(1 << 30)), extent=float(rng.choice([1e2, 1e4, 1e6])))
        knobs = dict(lanes=int(rng.choice([0, 2, 4, 8])), w=int(rng.choice([0, 4, 8, 16])), fused_chain=int(rng.choice([0, 1, 2])))
        dt = float(rng.choice([0.01, 0.005, 0.02]))
        steps = int(rng.choice([1, 2, 3, 5]))
        try:
            one = T.run(part, m, 1, dt, **knobs)
            if m:
                T.check_one_step(one, part, m, dt)
            a = T.run(part, m, steps, dt, graph=1, **knobs); b = T.run(part, m, steps, dt, graph=0, **knobs)
            assert a.tobytes() == b.tobytes()
            if steps > 1 and m:
                want = ob.step(part, m, dt, steps, kind="avx")
                d = T.rel_displacement(a, want, part)
                assert d <= 1e-4 or not np.isfinite(d), d
        except AssertionError as e:
            bad += 1
            print("FAIL", block, case, n, m, knobs, steps, str(e)[:200], flush=True)
print("done, failures:", bad)
