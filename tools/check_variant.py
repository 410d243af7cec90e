#!/usr/bin/env python3
"""Quick correctness check of one libnbody_hip.so (NBODY_HIP_SO) against the oracle: fixtures, several shapes."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb, oracle_binding as ob
ic = np.fromfile(os.path.join(ROOT, "tests/golden/ic_4096.bin"), dtype=np.float32).reshape(-1, 8)
part, m = ob.partition(ic)
acc64, mag = ob.acc_f64(part, m)
bound = 1e-4 * np.abs(acc64) + 1e-6 * mag
bad = 0
ref = None
for variant in (0, 1):
    for (k, w) in ((1, 1), (1, 16), (2, 4), (2, 16), (2, 1), (1, 8)):
        sim = nb.SimPipeline(4096, m); sim.configure(variant=variant, k=k, w=w); sim.set_data(part); sim.update(1, 0.01)
        got = sim.get_data(); sim.close()
        ratio = float(np.max(np.abs(got[:, 4:6] - acc64) / bound))
        ok = ratio <= 1.0
        bad += (not ok)
        print(f"{os.environ.get('NBODY_HIP_SO','default')[-30:]} variant={variant} k={k} w={w}: worst err/bound {ratio:.3g} {'ok' if ok else 'FAIL'}", flush=True)
sys.exit(1 if bad else 0)
