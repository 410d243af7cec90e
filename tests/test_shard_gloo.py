"""N > 1 host logic on the CPU: world_size-2 (and 3) gloo run of the shard plan + per-step all-gather."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world,n", [(2, 1024), (3, 333)])
def test_sharded_exchange_reproduces_single_rank(tmp_path, golden, world, n):
    steps, dt = 3, 0.01
    out = tmp_path / "sharded.bin"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + world + n % 97),
           os.path.join(HERE, "_gloo_worker.py"), str(out), str(n), str(steps), str(dt)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.fromfile(out, dtype=np.float32).reshape(-1, 8)
    part, m = ob.partition(golden(f"ic_{n}.bin"))
    want = ob.step(part, m, dt, steps)
    # pos/vel/mass/radius of every particle and acc of everything: bit-exact with the unsharded run
    assert got.tobytes() == want.tobytes()
