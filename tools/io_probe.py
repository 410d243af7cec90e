#!/usr/bin/env python3
"""Host<->device hand-over cost: SetSimulationData / GetSimulationData at several N."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb
for n in (6000, 65536, 1 << 20, 1 << 22):
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic); part = w.particles(); w.close()
    m = int((part[:, 6] > 0).sum())
    sim = nb.SimPipeline(n, m); sim.set_data(part); sim.update(1, 0.01); sim.get_data()
    t0 = time.perf_counter()
    for _ in range(5): sim.set_data(part)
    t1 = time.perf_counter()
    for _ in range(5): out = sim.get_data()
    t2 = time.perf_counter()
    sim.close()
    mb = n * 32 / 1e6
    print(f"N={n:8d} ({mb:7.1f} MB): Set {1e3*(t1-t0)/5:8.3f} ms ({mb/((t1-t0)/5)/1e3:5.1f} GB/s)  Get {1e3*(t2-t1)/5:8.3f} ms ({mb/((t2-t1)/5)/1e3:5.1f} GB/s)", flush=True)

# the World path: the particle array is page-locked by the pipeline (nb_hip_note_host_array)
import ctypes as C
for n in (65536, 1 << 20):
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic)
    w.update_gpu(0.01, 1); w.particles()
    L = nb.nbody_lib()
    t0 = time.perf_counter()
    for _ in range(5):
        w.update_gpu(0.01, 1)
    t1 = time.perf_counter()
    for _ in range(5):
        w.update_gpu(0.01, 1)
        cnt = C.c_uint32(); L.GetWorldParticles(w._h, C.byref(cnt))      # D2H into the World's own array, no numpy copy
    t2 = time.perf_counter()
    for _ in range(5):
        w.update_cpu(0.01, 0)                                            # marks the array dirty -> next GPU step uploads
        w.update_gpu(0.01, 1)
        cnt = C.c_uint32(); L.GetWorldParticles(w._h, C.byref(cnt))
    t3 = time.perf_counter()
    w.close()
    print(f"World N={n}: step {1e3*(t1-t0)/5:.2f} ms | step + GetWorldParticles {1e3*(t2-t1)/5:.2f} ms | upload + step + download {1e3*(t3-t2)/5:.2f} ms", flush=True)
