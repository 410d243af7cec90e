#!/usr/bin/env python3
"""Scratch probe run on the GPU box: parity on the 4096 fixture + a shape sweep. Not a test."""
import os, sys, time, json
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb
import oracle_binding as ob

print("device:", nb.device_info(), flush=True)
ic = np.fromfile(os.path.join(ROOT, "tests/golden/ic_4096.bin"), dtype=np.float32).reshape(-1, 8)
part, m = ob.partition(ic)
acc64, mag = ob.acc_f64(part, m)
ref1 = ob.step(part, m, 0.01, 1)
ref10 = ob.step(part, m, 0.01, 10)
for variant in (0, 1):
    for (k, w) in ((0, 0), (1, 1), (2, 4), (4, 4), (4, 16), (1, 16)):
        sim = nb.SimPipeline(part.shape[0], m)
        sim.configure(variant=variant, k=k, w=w)
        sim.set_data(part)
        sim.update(1, 0.01)
        out = sim.get_data()
        err = np.abs(out[:, 4:6] - acc64)
        bound = 1e-4 * np.abs(acc64) + 1e-6 * mag
        ok = bool(np.all(err <= bound))
        rel = float(np.max(err / (np.abs(acc64) + 1e-30)))
        referr = float(np.max(np.abs(ref1[:, 4:6] - acc64) / (np.abs(acc64) + 1e-30)))
        dpos = float(np.max(np.abs(out[:, 0:2] - ref1[:, 0:2])))
        sim.set_data(part)
        sim.update(10, 0.01)
        out10 = sim.get_data()
        l2 = float(np.linalg.norm(out10[:, 0:2].astype(np.float64) - ref10[:, 0:2]) / np.linalg.norm(ref10[:, 0:2].astype(np.float64)))
        print(f"variant={variant} k={k} w={w} shape={sim.launch_shape()} acc_ok={ok} max_rel_acc_err={rel:.2e} (avx path {referr:.2e}) dpos1={dpos:.2e} relL2_pos10={l2:.2e}", flush=True)
        sim.close()

def synth(n, seed=3):
    rng = np.random.default_rng(seed)
    a = np.zeros((n, 8), dtype=np.float32)
    r = 5e4 * np.sqrt(n / 65536.0) * np.sqrt(rng.random(n)); t = rng.random(n) * 2 * np.pi
    a[:, 0] = r * np.cos(t); a[:, 1] = r * np.sin(t)
    a[:, 2:4] = rng.standard_normal((n, 2)) * 10
    massive = rng.random(n) < 0.5
    a[:, 7] = np.where(massive, 1.5 + 8 * rng.random(n), 0.5)
    a[:, 6] = np.where(massive, 41.9 * a[:, 7] ** 3, 0.0)
    return ob.partition(a)

for n in (65536, 1 << 20):
    part, m = synth(n)
    print(f"N={n} M={m}", flush=True)
    for variant in (0, 1):
        for k in (1, 2, 4):
            for w in (1, 2, 4, 8, 16):
                groups = (n + 64 * k - 1) // (64 * k)
                if groups * w < 1024 or groups * w > 70000:
                    continue
                sim = nb.SimPipeline(n, m)
                sim.configure(variant=variant, k=k, w=w)
                sim.set_data(part)
                steps = 3 if n > 100000 else 20
                sim.update(1, 0.01)
                sim.update(steps, 0.01)
                ms, launches = sim.last_step_ms()
                per = ms / launches
                print(f"  variant={variant} k={k} w={w} wg={groups:6d} {per:9.3f} ms/step  {n * m / (per * 1e-3):.3e} int/s", flush=True)
                sim.close()
