#!/usr/bin/env python3
"""Driver for tools/profile_persist.sh: 100 plain-launch steps at size N with the auto shape, classic (persist = 1) or
persistent (persist = P) -- one process per setting so that a profiler's per-kernel rows belong to one setting."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb   # noqa: E402

assert nb.hip_lib().nb_hip_tuning_build(), "needs a `make -C nbody_amd/csrc TUNING=1` build (the persistent kernels are not in the shipped library)"

n, persist = int(sys.argv[1]), int(sys.argv[2])
ic = nb.make_galaxies(n, 2, seed=11037)
w = nb.World(ic)
part = w.particles()
w.close()
m = int((part[:, 6] > 0).sum())
probe = nb.SimPipeline(n, m)
probe.set_data(part)
probe.update(1, 0.01)
shape = probe.launch_shape()
probe.close()
sim = nb.SimPipeline(n, m)
sim.configure(graph=0, k=shape["k"], w=shape["w"], split=shape["split"], unit=shape["unit"], persist=persist)
sim.set_data(part)
sim.update(10, 0.01)
sim.update(100, 0.01)
print(f"N={n} M={m} persist={persist} shape={sim.launch_shape()}", flush=True)
sim.close()
