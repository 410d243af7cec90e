import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nbody_amd as nb
n = 1 << 20
ic = nb.make_galaxies(n, 2, seed=11037)
w = nb.World(ic)
w.update_gpu(0.01, 1); w.particles()
L = nb.nbody_lib()
def t(f, reps=5):
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return 1e3 * (time.perf_counter() - t0) / reps
def get():
    cnt = C.c_uint32(); L.GetWorldParticles(w._h, C.byref(cnt))
step = t(lambda: w.update_gpu(0.01, 1))
def sg(): w.update_gpu(0.01, 1); get()
def usd(): w.update_cpu(0.01, 0); w.update_gpu(0.01, 1); get()
def us(): w.update_cpu(0.01, 0); w.update_gpu(0.01, 1)
print(os.environ.get("TAG",""), f"step {step:.2f} | step+get {t(sg):.2f} | upload+step+get {t(usd):.2f} | upload+step {t(us):.2f} | step+get again {t(sg):.2f}")
