"""Which GPU work in this process disturbs libc's rand() stream?  (nbody-bench draws its universes from it, like the
reference's bench.c:42,53.)  Before every stage srand(1); after it the next 5 values are compared with the undisturbed
ones."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_amd as nb

libc = C.CDLL(None)
libc.srand(1)
plain = [libc.rand() for _ in range(5)]
ic = nb.make_galaxies(4096, 2, own_rng=True, seed=7)
w = nb.World(ic)
part = w.particles()
w.close()
m = int((part[:, 6] > 0).sum())


def stage(name, fn):
    libc.srand(1)
    out = fn()
    got = [libc.rand() for _ in range(5)]
    print(f"{name:34s}: {'same' if got == plain else 'DISTURBED'}")
    return out


stage("nb_hip_device_count", nb.device_count)
stage("nb_hip_device_info (first touch)", nb.device_info)
sim = stage("CreateSimPipeline", lambda: nb.SimPipeline(4096, m))
stage("SetSimulationData (first)", lambda: sim.set_data(part))
stage("PerformSimUpdate(1) (first launch)", lambda: sim.update(1, 0.01))
stage("PerformSimUpdate(40) (graph)", lambda: sim.update(40, 0.01))
stage("GetSimulationData", sim.get_data)
stage("DestroySimPipeline", sim.close)
sim2 = nb.SimPipeline(4096, m)
stage("second pipeline: Set + 40 steps", lambda: (sim2.set_data(part), sim2.update(40, 0.01)))
sim2.close()
