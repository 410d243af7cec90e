#!/usr/bin/env python3
"""The reference GUI's call pattern (src/main.c:157-163,237-250): every frame UpdateWorld_GPU(world, PHYS_STEP, updates)
then GetWorldParticles for drawing -- 6000 particles, 3 galaxies (main.c:13,44), `updates` = 1, 2, 4, 8 (the STEPS[]
multiplier).  Wall time per frame through the include/nbody.h surface, hipGraph policy 0 / 1 / 2, and the same without
the read-back.  Knobs are SET on the World's pipeline and READ BACK (NB_FRAME_KNOBS="readback=0,timing=1,zero_copy_upload=0":
nb_hip_configure / nb_hip_tune through World.tune) -- the shipped library reads no NB_HIP_READBACK / NB_HIP_TIMING /
NB_HIP_ZERO_COPY_UPLOAD from the environment (TUNING=1 builds only), so a label taken from those variables would describe a run
that never happened.  NB_HIP_GRAPH is a public preset and stays an environment variable."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import nbody_amd as nb
frames = 300
knobs = dict((k, int(v)) for k, v in (kv.split("=") for kv in os.environ.get("NB_FRAME_KNOBS", "").split(",") if kv))
sizes = [int(x) for x in sys.argv[1:]] or [6000, 1000, 20000]
for n in sizes:
    ic = nb.make_galaxies(n, 3, seed=11037)
    for updates in [int(x) for x in os.environ.get("NB_FRAME_UPDATES", "1,2,4,8").split(",")]:
        row = []
        for graph in (os.environ.get("NB_HIP_GRAPH", "2"),):   # read by CreateSimPipeline (the World owns its pipeline)
            w = nb.World(ic)
            if knobs:
                w.tune(**knobs)
                assert w.tune(**knobs) == knobs, "a knob did not take"     # the second call returns what the first one set
            for _ in range(5):
                w.update_gpu(0.01, updates); w.particles()
            t0 = time.perf_counter()
            for _ in range(frames):
                w.update_gpu(0.01, updates); w.particles()
            full = (time.perf_counter() - t0) / frames * 1e6
            t0 = time.perf_counter()
            for _ in range(frames):
                w.update_gpu(0.01, updates)
            bare = (time.perf_counter() - t0) / frames * 1e6
            t0 = time.perf_counter()
            for _ in range(frames):
                w.update_cpu(0.01, 0); w.update_gpu(0.01, updates)     # CPU touched the array: re-upload every frame
            mixed = (time.perf_counter() - t0) / frames * 1e6
            w.close()
            row.append(f"graph={graph}: {full:7.1f} us/frame ({bare:6.1f} without read-back, {mixed:6.1f} with re-upload + read-back)")
        tag = " ".join(f"{k}={v}" for k, v in knobs.items())
        print(f"[{tag or 'defaults'}] N={n:6d} updates={updates}: " + " | ".join(row), flush=True)
