import sys, time
sys.path.insert(0, '/root/repo')
import nbody_amd as nb
assert nb.hip_lib().nb_hip_tuning_build(), "needs a `make -C nbody_amd/csrc TUNING=1` build (the persistent kernels are not in the shipped library)"
import numpy as np
n = 1 << 20
ic = nb.make_galaxies(n, 2, seed=11037)
w = nb.World(ic); part = w.particles(); w.close()
m = int((part[:, 6] > 0).sum())
base = None
for persist in (0, 2, 8, 32, 0, 32):
    sim = nb.SimPipeline(n, m)
    sim.configure(graph=0, persist=persist, **({} if persist == 0 else dict(k=2, w=16, split=2, unit=64)))
    sim.set_data(part)
    sim.update(3, 0.01)
    t0 = time.perf_counter(); sim.update(10, 0.01); dt = (time.perf_counter() - t0) / 10
    ms, launches = sim.last_step_ms()
    out = sim.get_data()
    if base is None: base = out
    print(f"persist={persist:2d} shape={sim.launch_shape()} {dt*1e3:8.3f} ms/step kernel {ms/launches:7.3f} ms/launch bits {'same' if out.tobytes()==base.tobytes() else 'DIFFER'}", flush=True)
    sim.close()
