#!/usr/bin/env python3
"""Error of the GPU step vs float64 next to the reference AVX path's own error, at BASELINE sizes (sampled)."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb
import oracle_binding as ob
for n in [int(x) for x in sys.argv[1:]] or (4096, 65536, 262144, 1 << 20):
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic); part = w.particles(); w.close()
    m = int((part[:, 6] > 0).sum())
    rng = np.random.default_rng(n)
    idx = (np.arange(n) if n <= 2000 else np.unique(rng.integers(0, n, 2000))).astype(np.uint32)
    acc64, mag = ob.acc_f64_subset(part, m, idx)
    avx = ob.acc_avx_subset(part, m, idx).astype(np.float64)
    e_avx = np.abs(avx - acc64)
    # every knob on auto (lane-split below N x M = 9e6), the classic auto shape, one wave per tile, the LDS-tile route
    for knobs in (dict(), dict(lanes=1), dict(w=1, k=1), dict(variant=0)):
        sim = nb.SimPipeline(n, m); sim.configure(**knobs); sim.set_data(part); sim.update(1, 0.01)
        got = sim.get_data()[idx, 4:6].astype(np.float64); shape = sim.launch_shape(); sim.close()
        e = np.abs(got - acc64)
        b1 = 1e-4 * np.abs(acc64) + 1e-6 * mag
        print(f"N={n} M={m} {shape}: GPU max err/bound {np.max(e/b1):7.3f}  AVX max err/bound {np.max(e_avx/b1):7.3f} | "
              f"rms(err/mag) GPU {np.sqrt(np.mean((e/mag)**2)):.2e} AVX {np.sqrt(np.mean((e_avx/mag)**2)):.2e} | "
              f"max(err/mag) GPU {np.max(e/mag):.2e} AVX {np.max(e_avx/mag):.2e} | max rel GPU {np.max(e/np.abs(acc64)):.2e} AVX {np.max(e_avx/np.abs(acc64)):.2e}", flush=True)
