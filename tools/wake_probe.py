#!/usr/bin/env python3
"""How much load does the GPU need before the reference harness' first timed call runs at the clock the chip holds?

The reference nbody-bench creates a World, runs 10 warm-up steps and times ONE 100-step call (src/bench.c:21-35).  After
host-side work the GPU sits at an idle clock and the 10 warm-up steps (0.2-3 ms at N = 10 000 ... 50 000) do not bring it
back: the timed call runs 5-20 % slower than the same call repeated (profiles/r02_warmup_ramp.txt).  This probe idles the
GPU (sleep), uploads a world, burns T ms of the clock probe's loop (nb_hip_probe_clock) as a stand-in for a "wake" at
upload, then runs the harness' sequence and prints us/step of the first timed call and of the fifth."""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb   # noqa: E402


def universe(n):
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic)
    part = w.particles()
    w.close()
    return part, int((part[:, 6] > 0).sum())


idle_s = float(os.environ.get("IDLE_S", "1.5"))
for n in [int(x) for x in sys.argv[1:]] or (10000, 20000, 50000):
    part, m = universe(n)
    for wake_ms in (0, 2, 5, 10, 20, 40, 80, 0):
        time.sleep(idle_s)                      # host-side work of the harness (universe, CPU column): the GPU idles
        sim = nb.SimPipeline(n, m)
        sim.set_data(part)
        if wake_ms:
            nb.probe_clock(float(wake_ms))
        sim.update(10, 1.0)
        calls = []
        for _ in range(5):
            t0 = time.perf_counter()
            sim.update(100, 1.0)
            calls.append((time.perf_counter() - t0) / 100 * 1e6)
        sim.close()
        print(f"N={n:6d} wake {wake_ms:3d} ms: first timed call {calls[0]:8.2f} us/step, fifth {calls[4]:8.2f}  ({calls[0] / calls[4] - 1:+.1%})", flush=True)
