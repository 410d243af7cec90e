#!/usr/bin/env python3
"""Auto launch shape vs unsplit K=2/W=16 and K=1/W=16 over a range of N (tuning aid for choose_shape)."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb
for n in [int(x) for x in sys.argv[1:]] or [4096, 16384, 20000, 50000, 65536, 100000, 200000, 300000, 1000000]:
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic); part = w.particles(); w.close()
    m = int((part[:, 6] > 0).sum())
    row = []
    for knobs in (dict(), dict(k=2, w=16, split=1), dict(k=1, w=16, split=1)):
        sim = nb.SimPipeline(n, m); sim.configure(**knobs); sim.set_data(part)
        steps = 4 if n > 300000 else (30 if n > 30000 else 200)
        sim.update(3, 0.01)
        best = 1e9
        for _ in range(3):
            sim.update(steps, 0.01); ms, _l = sim.last_step_ms(); best = min(best, ms / steps)
        sh = sim.launch_shape(); sim.close()
        row.append((best, sh))
    print(f"N={n:8d} M={m:7d}: auto {row[0][0]*1e3:9.1f} us {n*m/row[0][0]/1e-3:.3e} int/s k={row[0][1]['k']} w={row[0][1]['w']} split={row[0][1]['split']:2d} wg={row[0][1]['workgroups']:6d} | "
          f"k2w16 {row[1][0]*1e3:9.1f} us | k1w16 {row[2][0]*1e3:9.1f} us | auto/best-unsplit {row[0][0]/min(row[1][0], row[2][0]):.3f}", flush=True)
