#!/usr/bin/env python3
"""How far the GPU path and the reference AVX path are from EACH OTHER at the BASELINE sizes.

north_star: "results match the reference AVX CPU path ... within a stated fp32 tolerance" (reference
src/lib/sim_cpu.c:156-194 driven as world.c:99-110).  Both paths evaluate the same fp32 terms; they differ in the order
the M terms of a receiver's sum are added (AVX: eight sequential lane sums of M/8 terms each, then the lanes; GPU: per
wave-slice Kahan-free partial sums combined per workgroup), so the deviation scales with sum_j |contribution_j| and
grows with M on the AVX side (a sequential fp32 sum of M/8 terms).  For every size this prints
  one step, sampled receivers:  max and rms of |acc_gpu - acc_avx| / sum_j |contribution_j|, and the same of each path
                                against the float64 sum (which of the two carries the difference);
  K steps, all particles:       rel_displacement = |dpos_gpu - dpos_avx| / |dpos_avx| and relative L2 of the velocities
                                against the bit-exact AVX restatement stepped K times on the host cores.
tests/test_gpu_parity.py::test_gpu_versus_the_avx_path_at_every_baseline_size (constants: tests/gpu_common.py GPU_VS_AVX) asserts the constants DESIGN.md section 5
states from this table.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd as nb          # noqa: E402
import oracle_binding as ob     # noqa: E402

# (N, steps of the multi-step comparison): ten steps where the host cores finish them in seconds, two at 2^20
CASES = ((65536, 10), (262144, 10), (1 << 20, 2))
DT = 0.01


def universe(n):
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic)
    part = w.particles()
    w.close()
    return part, int((part[:, 6] > 0).sum())


def measure(n, steps, samples=2000, **knobs):
    part, m = universe(n)
    rng = np.random.default_rng(n)
    idx = np.unique(np.concatenate([[0, 1, m - 1, m, n - 1], rng.integers(0, n, samples)])).astype(np.uint32)
    acc64, mag = ob.acc_f64_subset(part, m, idx)
    avx = ob.acc_avx_subset(part, m, idx).astype(np.float64)
    sim = nb.SimPipeline(n, m)
    sim.configure(**knobs)
    sim.set_data(part)
    sim.update(1, DT)
    one = sim.get_data()
    shape = sim.launch_shape()
    gpu = one[idx, 4:6].astype(np.float64)
    # every further step is tie-broken against float64 too, from the GPU's OWN state at the start of the step:
    # the sampled accelerations the step stored must be closer to the float64 sum of that state than the AVX order's are
    tie = []
    state = one
    for _ in range(steps - 1):
        sim.update(1, DT)
        nxt = sim.get_data()
        a64, mg = ob.acc_f64_subset(state, m, idx)
        a_avx = ob.acc_avx_subset(state, m, idx).astype(np.float64)
        g = float((np.abs(nxt[idx, 4:6].astype(np.float64) - a64) / mg).max())
        a = float((np.abs(a_avx - a64) / mg).max())
        tie.append((g, a))
        state = nxt
    got = state
    sim.close()
    t0 = time.perf_counter()
    want = ob.step(part, m, DT, steps, kind="avx")
    cpu_s = time.perf_counter() - t0
    p0 = part[:, 0:2].astype(np.float64)
    dg, dw = got[:, 0:2].astype(np.float64) - p0, want[:, 0:2].astype(np.float64) - p0
    vw = want[:, 2:4].astype(np.float64)

    def stats(a, b):
        r = np.abs(a - b) / mag
        return float(r.max()), float(np.sqrt(np.mean(r ** 2)))

    out = {"n": n, "m": m, "steps": steps, "samples": int(idx.size), "shape": shape, "cpu_seconds": cpu_s}
    out["gpu_avx_max"], out["gpu_avx_rms"] = stats(gpu, avx)
    out["gpu_f64_max"], out["gpu_f64_rms"] = stats(gpu, acc64)
    out["avx_f64_max"], out["avx_f64_rms"] = stats(avx, acc64)
    out["rel_displacement"] = float(np.linalg.norm(dg - dw) / np.linalg.norm(dw))
    out["rel_l2_vel"] = float(np.linalg.norm(got[:, 2:4].astype(np.float64) - vw) / np.linalg.norm(vw))
    out["rel_l2_pos"] = float(np.linalg.norm(got[:, 0:2].astype(np.float64) - want[:, 0:2]) / np.linalg.norm(want[:, 0:2].astype(np.float64)))
    out["static_equal"] = bool(np.array_equal(got[:, 6:8], want[:, 6:8]))
    # later steps: worst GPU-f64 and the AVX-f64 of the same step, as fractions of sum|contrib| (empty when steps == 1)
    out["later_steps_gpu_f64_max"] = max((g for g, _ in tie), default=0.0)
    out["later_steps_avx_f64_min"] = min((a for _, a in tie), default=0.0)
    out["later_steps_tie_break_holds"] = all(g <= a for g, a in tie)
    return out


def line(r):
    return (f"N={r['n']} M={r['m']} {r['shape']}\n"
            f"  one step, {r['samples']} receivers, |d acc| / sum|contrib|: GPU-AVX max {r['gpu_avx_max']:.2e} rms {r['gpu_avx_rms']:.2e} | "
            f"GPU-f64 max {r['gpu_f64_max']:.2e} rms {r['gpu_f64_rms']:.2e} | AVX-f64 max {r['avx_f64_max']:.2e} rms {r['avx_f64_rms']:.2e}\n"
            f"  {r['steps']} steps at dt={DT} vs the AVX stepper ({r['cpu_seconds']:.1f} s of host cores): rel_displacement {r['rel_displacement']:.2e}, "
            f"rel L2 vel {r['rel_l2_vel']:.2e}, rel L2 pos {r['rel_l2_pos']:.2e}, mass/radius equal {r['static_equal']}\n"
            f"  float64 tie-break at every later step, from the GPU's own state: worst GPU-f64 {r['later_steps_gpu_f64_max']:.2e} vs smallest "
            f"AVX-f64 {r['later_steps_avx_f64_min']:.2e} of sum|contrib| -> {'holds' if r['later_steps_tie_break_holds'] else 'FAILS'}")


if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1:]]
    for n, steps in CASES:
        if sizes and n not in sizes:
            continue
        for knobs in (dict(), dict(variant=0)):
            print(line(measure(n, steps, **knobs)), flush=True)
