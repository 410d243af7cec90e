"""The harnesses on the GPU (needs an MI355X: `pytest -m gpu`): the reference's own world.c / bench.c on top of libnbody_hip.so, the
product's nbody-bench (table, --verify, --gpus P with its transport chain), the library's watchdog and rand() hygiene, the measurement
aids, and the single-GPU bench line (tolerances: tests/test_gpu_parity.py docstring; helpers: tests/gpu_common.py).
"""
import ctypes as C  # noqa: F401
import os  # noqa: F401
import subprocess  # noqa: F401
import sys  # noqa: F401
import time  # noqa: F401

import numpy as np  # noqa: F401
import pytest

import nbody_amd as nb  # noqa: F401
import oracle_binding as ob  # noqa: F401
from gpu_common import *  # noqa: F401,F403  -- helpers shared by the GPU test files (tests/gpu_common.py)

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------------------------
# drop-in: the reference's own world.c / bench.c on top of libnbody_hip.so (oracle/_ref travels prebuilt)
# ---------------------------------------------------------------------------------------------------------------

@pytest.mark.skipif(not os.path.exists(ob.REF_WORLD_SO), reason="oracle/_ref/libnbody_ref_world.so not built")
def test_reference_world_c_drives_our_hip_pipeline(golden):
    nb.hip_lib()
    ref = C.CDLL(ob.REF_WORLD_SO)
    ref.CreateWorld.restype = C.c_void_p
    ref.CreateWorld.argtypes = [C.c_void_p, C.c_uint32]
    ref.GetWorldParticles.restype = C.c_void_p
    ref.GetWorldParticles.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    ref.UpdateWorld_GPU.argtypes = [C.c_void_p, C.c_float, C.c_uint32]
    ref.UpdateWorld_CPU.argtypes = [C.c_void_p, C.c_float, C.c_uint32]
    ref.DestroyWorld.argtypes = [C.c_void_p]
    ic = golden("ic_1024.bin")
    w = ref.CreateWorld(ic.ctypes.data, 1024)
    ref.UpdateWorld_GPU(w, 0.01, 2)
    ref.UpdateWorld_CPU(w, 0.01, 1)
    ref.UpdateWorld_GPU(w, 0.01, 1)
    n = C.c_uint32()
    p = ref.GetWorldParticles(w, C.byref(n))
    got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n.value, 8)).copy()
    ref.DestroyWorld(w)
    ours = nb.World(ic)
    ours.update_gpu(0.01, 2)
    ours.update_cpu(0.01, 1)
    ours.update_gpu(0.01, 1)
    want = ours.particles()
    ours.close()
    assert got.tobytes() == want.tobytes()


def test_nbody_bench_gpu_column():
    """nbody-bench --gpu: the GPU column alone (reference src/bench.c:41-74 with --gpu), and -- with --verify 5 -- what that
    column computed: 5 steps of UpdateWorld_GPU against 5 steps of UpdateWorld_CPU per row, printed and asserted."""
    import re
    exe = os.path.join(nb.LIB_DIR, "nbody-bench")
    r = subprocess.run([exe, "--gpu", "--n", "4000", "--n", "20000", "--steps", "10", "--warmup", "2", "--dt", "0.01", "--verify", "5"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l.split() for l in r.stdout.strip().splitlines()]
    assert rows[0][:2] == ["N", "GPU"] and [x[0] for x in rows[1:]] == ["4000", "20000"]
    assert all(float(x[2]) > 1e9 for x in rows[1:])
    # us/step x interactions/s = N x M of the row: the two printed figures describe the same run
    for x, n in zip(rows[1:], (4000, 20000)):
        pairs = float(x[1]) * 1e-6 * float(x[2])
        assert 0.3 * n * n <= pairs <= 0.7 * n * n, (x, pairs)      # M ~ N / 2 with galaxy.h ICs
    devs = [float(v) for v in re.findall(r"GPU vs CPU rel_displacement ([0-9.e+-]+)", r.stderr)]
    assert len(devs) == 2 and all(d <= 1e-5 for d in devs), r.stderr      # stated tolerance 1e-4; observed ~1e-6
    assert r.stderr.count("mass/radius equal yes") == 2


def test_nbody_bench_verify_column_compares_gpu_with_the_cpu_path():
    """nbody-bench --verify K: K steps of UpdateWorld_GPU against K steps of UpdateWorld_CPU (bit-exact with the reference
    AVX build, tests/test_world_cpu.py) per row, relative to what the steps moved; the harness itself fails above 1e-4."""
    import re
    exe = os.path.join(nb.LIB_DIR, "nbody-bench")
    r = subprocess.run([exe, "--n", "1200", "--n", "10000", "--steps", "5", "--warmup", "1", "--dt", "0.01", "--verify", "10"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    devs = [float(x) for x in re.findall(r"GPU vs CPU rel_displacement ([0-9.e+-]+)", r.stderr)]
    assert len(devs) == 2 and all(d <= 1e-5 for d in devs), r.stderr      # observed ~1e-6: summation order only
    assert r.stderr.count("mass/radius equal yes") == 2


def test_rccl_watchdog_arms_per_wait_and_fires_on_a_wait_that_overruns(golden):
    """The blocking waits of an RCCL pipeline run under ONE long-lived watcher thread that is armed with a deadline and
    disarmed again (rccl_bind.hip): 300 short blocking calls arm / disarm it without tripping, and a chain that outlasts
    NB_HIP_COMM_TIMEOUT_S ends the process with the diagnostic and exit code 3 -- no retry, no re-exec."""
    code = ("import os, sys, numpy as np, nbody_amd as nb\n"
            "ic = nb.make_galaxies(65536, 2, own_rng=True, seed=5)\n"
            "w = nb.World(ic); part = w.particles(); w.close(); m = int((part[:, 6] > 0).sum())\n"
            "sim = nb.SimPipeline(65536, m, rank=0, nranks=1, unique_id=nb.comm_unique_id())\n"
            "assert sim.comm_info()['owns_comm']\n"
            "sim.set_data(part)\n"
            "os.environ['NB_HIP_COMM_TIMEOUT_S'] = '2'   # read at every wait; the communicator's creation had the default\n"
            "for _ in range(300): sim.update(1, 0.01)\n"
            "print('SHORT CALLS OK', flush=True)\n"
            "sim.update(int(sys.argv[1]), 0.01)\n"
            "print('LONG CALL RETURNED', flush=True)\n")
    env = dict(os.environ, NB_HIP_FORCE_SHARDED="1")
    env.pop("NB_HIP_COMM_TIMEOUT_S", None)
    ok = subprocess.run([sys.executable, "-c", code, "500"], cwd=nb.ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0 and "LONG CALL RETURNED" in ok.stdout, (ok.stdout, ok.stderr[-2000:])   # ~0.25 s: inside the bound
    late = subprocess.run([sys.executable, "-c", code, "12000"], cwd=nb.ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert late.returncode == 3, (late.returncode, late.stdout, late.stderr[-2000:])                    # ~5 s of steps against 2 s
    assert "SHORT CALLS OK" in late.stdout and "LONG CALL RETURNED" not in late.stdout
    assert "[watchdog] rank 0 of 1" in late.stderr and "did not complete within 2 s" in late.stderr


def test_explicit_lanes_or_route_keeps_the_per_step_kernel_on_tiny_worlds():
    """"fused_chain" auto applies only while the launch shape is on auto -- an explicit lanes / variant asks for
    the per-step kernel also on a world small enough for the one-workgroup chain."""
    part, m = synth(250, 0.5, seed=11)
    want = run(part, m, 10, 0.01, fused_chain=0, lanes=4)
    for knobs, fused in ((dict(), 10), (dict(lanes=4), 0), (dict(variant=0), 0), (dict(lanes=1), 0)):
        sim = nb.SimPipeline(250, m)
        sim.configure(**knobs)
        sim.set_data(part)
        sim.update(10, 0.01)
        assert sim.fused_steps() == fused, (knobs, sim.fused_steps())
        if knobs == dict(lanes=4):
            assert sim.launch_shape()["lanes"] == 4 and sim.get_data().tobytes() == want.tobytes()
        sim.close()


def test_gpu_work_leaves_the_callers_rand_stream_alone():
    """The reference harness seeds libc's rand() once and draws every universe of its table from it between GPU calls
    (src/bench.c:42,53); the HIP runtime's first set-up draws from the same process-global state.  The library swaps a
    private state in around it (RandGuard): in a fresh process, the values after srand(1) are the same with and
    without a World's whole GPU life in between."""
    code = ("import ctypes as C, numpy as np, nbody_amd as nb\n"
            "libc = C.CDLL(None); libc.srand(1); plain = [libc.rand() for _ in range(5)]\n"
            "ic = nb.make_galaxies(2000, 2, own_rng=True, seed=3)\n"
            "libc.srand(1)\n"
            "w = nb.World(ic); w.update_gpu(0.01, 3); w.particles(); w.update_gpu(0.01, 40); w.close()\n"
            "got = [libc.rand() for _ in range(5)]\n"
            "print('SAME' if got == plain else 'DISTURBED', plain, got)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=nb.ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("SAME"), (r.stdout, r.stderr[-2000:])


@pytest.mark.parametrize("transport", ["shm", "ipc"])
@pytest.mark.parametrize("P,n", [(2, 4000), (3, 1200), (3, 20000)])
def test_nbody_bench_c_ranks_on_one_gpu_bitwise(P, n, transport):
    """nbody-bench --gpus P --transport shm | ipc: P REAL processes forked by the C harness before anything touched HIP,
    each with its own HIP context on this one GPU, one World stepped through CreateWorldShardedWith over the shared page
    (shm: data staged through the host) or CreateWorldShardedDirect (ipc: every rank maps its peers' source arrays with
    hipIpcOpenMemHandle and pushes its slice into them device to device; the page carries handles and one barrier per
    step) -- no Python, no torch, /opt/rocm's HIP runtime.  With one wave per workgroup (--one-wave) the
    summation order does not depend on the launch geometry: the in-stream (plain) step must equal the single-GPU World
    bit for bit."""
    import re
    r = _bench_ranks(["--gpus", str(P), "--transport", transport, "--n", str(n), "--steps", "6", "--warmup", "2", "--dt", "0.01",
                      "--modes", "plain", "--verify", "4", "--one-wave"])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    v = re.findall(r"verify N=(\d+) mode=(\w+) steps=4: ranks agree (\w+); vs single GPU: rel_l2_pos ([0-9.e+-]+) max_abs_pos ([0-9.e+-]+) bitwise (\w+)", r.stderr)
    assert v == [(str(n), "plain", "yes", v[0][3], v[0][4], "yes")] and float(v[0][3]) == 0.0, r.stderr
    rows = [l.split() for l in r.stdout.strip().splitlines()]
    assert rows[0][:4] == ["N", "ranks", "mode", "GPU"] and rows[1][:3] == [str(n), str(P), "plain"]
    assert float(rows[1][3]) > 0 and float(rows[1][-2]) > 0 and float(rows[1][-1]) > 0     # us/step, kernel ms, gather ms
    assert f"{P} ranks, transport {transport}; ranks_with_communicator=0" in r.stderr
    assert ("direct device-to-device pushes" in r.stderr) == (transport == "ipc")


@pytest.mark.parametrize("transport", ["shm", "ipc"])
def test_nbody_bench_c_ranks_default_shapes_and_overlap(transport):
    """The same with the library's own launch shapes, both step modes, two sizes in one run (the second World gets a
    fresh exchange): every rank holds the same bytes and they stay within 1e-5 relative L2 of the single-GPU positions
    (the harness' own bound; observed ~1e-8)."""
    import re
    r = _bench_ranks(["--gpus", "2", "--transport", transport, "--n", "4096", "--n", "65536", "--steps", "5", "--warmup", "1", "--dt", "0.01"])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    v = re.findall(r"verify N=(\d+) mode=(\w+) steps=3: ranks agree (\w+); vs single GPU: rel_l2_pos ([0-9.e+-]+)", r.stderr)
    assert [(a, b, c) for a, b, c, _ in v] == [("4096", "plain", "yes"), ("4096", "overlap", "yes"), ("65536", "plain", "yes"),
                                               ("65536", "overlap", "yes")], r.stderr
    assert all(float(x[3]) <= 1e-6 for x in v)
    rows = [l.split() for l in r.stdout.strip().splitlines()][1:]
    assert [(x[0], x[2]) for x in rows] == [("4096", "plain"), ("4096", "overlap"), ("65536", "plain"), ("65536", "overlap")]
    assert all(float(x[5]) > 1e9 for x in rows)
    # --speedup: rank 0 times the same call on a single-GPU World; with every rank on ONE GPU the "speedup" is below 1
    r = _bench_ranks(["--gpus", "2", "--transport", transport, "--n", "20000", "--steps", "5", "--warmup", "1", "--dt", "0.01",
                      "--modes", "plain", "--verify", "0", "--speedup"])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    head, row = [l.split() for l in r.stdout.strip().splitlines()]
    assert head[-2:] == ["us", "speedup"] and 0.05 < float(row[-1]) < 1.5 and float(row[-2]) > 10
    # ... and a multi-rank row is never printed unchecked: --verify 0 is overridden (one step, ranks agree, = one GPU)
    assert "--verify 0 is not accepted with --gpus 2" in r.stderr and "verify N=20000 mode=plain steps=1: ranks agree yes" in r.stderr


def test_nbody_bench_c_one_forced_rccl_rank():
    """nbody-bench --gpus 1 --force-sharded: the RCCL path (ncclCommInitRank, in-place ncclAllGather per step, the chain
    captured as a hipGraph, the overlapped step) with ONE rank, forked by the C harness -- the HIP runtime and librccl
    this binds are /opt/rocm's (no torch in the process), which is what a real --gpus 8 run binds too."""
    import re
    r = _bench_ranks(["--gpus", "1", "--force-sharded", "--n", "20000", "--steps", "8", "--warmup", "2", "--dt", "0.01"])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "1 ranks, transport rccl; ranks_with_communicator=1 ncclCommCount=1..1" in r.stderr
    lib = re.search(r"lib=(\S+)", r.stderr).group(1)
    assert "librccl" in lib and "torch" not in lib, lib
    v = re.findall(r"mode=(\w+) steps=3: ranks agree yes; vs single GPU: rel_l2_pos ([0-9.e+-]+)", r.stderr)
    assert [m for m, _ in v] == ["plain", "overlap", "graph"] and all(float(x) <= 1e-6 for _, x in v), r.stderr
    rows = [l.split() for l in r.stdout.strip().splitlines() if l.split() and l.split()[0] == "20000"]
    assert [x[2] for x in rows] == ["plain", "overlap", "graph"] and all(float(x[5]) > 1e9 for x in rows)


def test_nbody_bench_c_ranks_refuses_rccl_without_enough_devices():
    if nb.device_count() >= 2:
        pytest.skip("more than one GPU here")
    r = _bench_ranks(["--gpus", "2", "--transport", "rccl", "--n", "1200", "--steps", "2"], timeout=120)
    assert r.returncode != 0 and "needs 2 (one per rank)" in r.stderr and "transport_fallback" not in r.stderr


def test_nbody_bench_c_falls_back_to_the_direct_exchange_in_fresh_ranks():
    """nbody-bench --gpus 2 with the default --transport auto on a one-GPU box: the RCCL attempt's ranks end with an error
    (two ranks, one device), the parent -- which never touches HIP -- forks a FRESH set of ranks over the direct exchange,
    says so on both streams, and the table of the second attempt is verified against a single-GPU World like any other
    (the C harness' half; reference shape: one plain command, src/bench.c:41-74)."""
    import re
    if nb.device_count() >= 2:
        pytest.skip("more than one GPU here: the RCCL attempt would succeed")
    r = _bench_ranks(["--gpus", "2", "--n", "20000", "--steps", "5", "--warmup", "1", "--dt", "0.01"], timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "transport_fallback rccl -> ipc" in r.stderr and "2 ranks, transport ipc" in r.stderr
    lines = r.stdout.strip().splitlines()
    mark = next(i for i, l in enumerate(lines) if l.startswith("# transport_fallback rccl -> ipc"))
    rows = [l.split() for l in lines[mark + 2:]]       # header, then one row per mode (plain, overlap: no captured graph over ipc)
    assert [(x[0], x[2]) for x in rows] == [("20000", "plain"), ("20000", "overlap")] and all(float(x[5]) > 1e9 for x in rows)
    v = re.findall(r"verify N=20000 mode=(\w+) steps=3: ranks agree (\w+); vs single GPU: rel_l2_pos ([0-9.e+-]+)", r.stderr)
    assert [(a, b) for a, b, _ in v] == [("plain", "yes"), ("overlap", "yes")] and all(float(x[2]) <= 1e-6 for x in v)
    # preflight: every rank of the ipc attempt wrote its device, peer row and one IPC open of the next rank
    pre = re.findall(r"# preflight rank (\d) of 2 transport=ipc device=0/1 pci=(\S+) can_access_peer=\[1\] ipc_export=0 ipc_open\(rank (\d)\)=0 ", r.stderr)
    assert sorted((a, c) for a, _, c in pre) == [("0", "1"), ("1", "0")] and len({b for _, b, _ in pre}) == 1, r.stderr[-3000:]


def test_nbody_bench_c_walks_to_shm_when_the_driver_refuses_ipc():
    """The C harness' whole chain, nothing rehearsed: with the IPC mode this pool's driver does not serve
    (HSA_ENABLE_IPC_MODE_LEGACY=1: hipIpcGetMemHandle -> invalid argument) `nbody-bench --gpus 2` goes rccl (status 2: one device
    for two ranks) -> ipc (status 134: abort() at the first IPC export, which the preflight line had already reported) -> shm,
    whose table is verified against a single-GPU World like any other (profiles/r06_legacy_ipc_cbench.txt is this run, kept)."""
    import re
    if nb.device_count() >= 2:
        pytest.skip("more than one GPU here: the RCCL attempt would succeed")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="1", OMP_NUM_THREADS="4")
    r = subprocess.run([os.path.join(nb.LIB_DIR, "nbody-bench"), "--gpus", "2", "--n", "65536", "--steps", "5", "--warmup", "1", "--dt", "0.01"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=nb.ROOT)
    marks = [l for l in r.stdout.splitlines() if l.startswith("# transport_fallback")]
    if len(marks) == 1 and r.returncode == 0:
        pytest.skip("this box's driver serves the legacy IPC mode: the direct exchange came up")
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert [m.split("(")[0].strip() for m in marks] == ["# transport_fallback rccl -> ipc", "# transport_fallback ipc -> shm"]
    assert "bring_up_failed, status 2" in marks[0] and "bring_up_failed, status 134" in marks[1]
    assert len(re.findall(r"# preflight rank \d of 2 transport=ipc .* ipc_export=[1-9]\d* \(", r.stderr)) == 2     # said why, before the abort
    assert len(re.findall(r"# preflight rank \d of 2 transport=shm .* ipc=not probed", r.stderr)) == 2
    rows = [l.split() for l in r.stdout.splitlines() if l.split() and l.split()[0] == "65536"]
    assert [(x[1], x[2]) for x in rows] == [("2", "plain"), ("2", "overlap")] and all(float(x[5]) > 1e11 for x in rows)
    v = re.findall(r"verify N=65536 mode=(\w+) steps=3: ranks agree (\w+); vs single GPU: rel_l2_pos ([0-9.e+-]+)", r.stderr)
    assert [(a, b) for a, b, _ in v] == [("plain", "yes"), ("overlap", "yes")] and all(float(x[2]) <= 1e-6 for x in v)
    assert "verification_failed" not in r.stdout + r.stderr


@pytest.mark.skipif(not os.path.exists(os.path.join(ob.ORACLE_DIR, "_ref", "nbody-bench-ref")),
                    reason="oracle/_ref/nbody-bench-ref not built")
def test_reference_bench_c_runs_unchanged_on_our_library():
    exe = os.path.join(ob.ORACLE_DIR, "_ref", "nbody-bench-ref")
    r = subprocess.run([exe, "--gpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l.split() for l in r.stdout.strip().splitlines()]
    assert rows[0] == ["N", "GPU"]
    assert [int(x[0]) for x in rows[1:]] == [250, 500, 800, 1200, 2000, 4000, 10000, 20000, 50000, 100000]


# ---------------------------------------------------------------------------------------------------------------
# measurement aids: the clock probe and the clock sampler (include/nbody_hip.h; bench.py roofline.held_clock_ghz)
# ---------------------------------------------------------------------------------------------------------------

def test_clock_sampler_runs_beside_the_step_kernels_without_touching_their_results():
    """nb_hip_clock_sampler_*: eight one-wave workgroups stamp the shader clock while a step chain runs on the pipeline's own
    stream.  The chain's results are bit-identical with and without the sampler, the sampler covers the chain's span, sits
    on several XCDs, and leaves by itself when its bound passes even if nobody stops it."""
    n = 65536
    _, part, m = bench_universe(n)
    want = run(part, m, 20, 0.01)
    sim = nb.SimPipeline(n, m)
    sim.set_data(part)
    assert nb.clock_sampler_begin(0.2, 4000.0) == 8
    sim.update(20, 0.01)
    got = sim.get_data()
    s = nb.clock_sampler_end()
    sim.close()
    assert got.tobytes() == want.tobytes()
    assert s["intervals"] >= 8 and 1.0 <= s["clock_ghz_min"] <= s["clock_ghz"] <= s["clock_ghz_max"] <= 2.45, s
    assert s["span_ms"] >= 5.0 and sum(1 for v in s["per_xcd_ghz"] if v > 0) >= 2, s
    assert all(1.0 <= v <= 2.45 for v in s["profile_ghz"]), s
    # bounded: never stopped from the host, the waves leave after max_ms by themselves (end() then only collects; how long
    # that takes is asked in tests/test_gpu_zz_perf.py)
    nb.clock_sampler_begin(0.2, 100.0)
    time.sleep(0.5)
    late = nb.clock_sampler_end()
    assert late["intervals"] >= 8 and late["span_ms"] >= 50.0, late


@pytest.mark.skipif(not nb.hip_lib().nb_hip_tuning_build(), reason="the persistent-launch experiment kernels are built with make TUNING=1 only")
def test_persistent_launch_equals_the_classic_launch():
    """The persistent-launch experiment kernel (tuning hook "persist"; closed: slower at every size,
    profiles/r05_persist_probe.txt): a launch of 1/P as many workgroups whose waves walk P (tile, part) work items each runs,
    per item, the code a classic workgroup runs -- same bits as the classic launch, with the finish kernel and with the fused
    finish, as plain launches and inside a hipGraph; and the classic launch of that state is what the oracle checks."""
    _, part, m = bench_universe(10000)
    base = run(part, m, 12, 0.01, graph=0)
    sim = nb.SimPipeline(10000, m)
    sim.set_data(part)
    sim.update(1, 0.01)
    shape = sim.launch_shape()
    one = sim.get_data()
    sim.close()
    check_one_step(one, part, m, 0.01)
    fixed = {k: shape[k] for k in ("k", "w", "split", "unit")}
    assert run(part, m, 1, 0.01, persist=2, **fixed).tobytes() == one.tobytes()
    assert shape["split"] > 1 and shape["lanes"] == 1
    for persist in (2, 3, 7):
        for fused in (0, 1):
            for graph in (0, 1):
                got = run(part, m, 12, 0.01, graph=graph, fused_finish=fused, persist=persist, **fixed)
                assert got.tobytes() == base.tobytes(), (persist, fused, graph)
    sim = nb.SimPipeline(10000, m)
    sim.configure(persist=2, **fixed)
    sim.set_data(part)
    sim.update(1, 0.01)
    assert sim.launch_shape()["workgroups"] == (shape["workgroups"] + 1) // 2
    sim.close()


@pytest.mark.parametrize("leg", ["clock probe", "repeats", "extra_configs C2/C3/N2/C1"])
def test_single_gpu_bench_line_survives_a_leg_that_aborts(leg):
    """The driver's own command (`python bench.py`, one GPU): every leg after the headline -- clock probe, parity stamp,
    repeats, the clock-sampler leg, the LDS route, extra_configs -- runs with the line in hand.  A leg that dies by the
    library's error convention (print + abort(), reference src/lib/util.h:17-29) still leaves the headline on stdout, once,
    with "extras_aborted" naming the leg, and the run does not report success."""
    import json
    env = dict(os.environ, NB_BENCH_REHEARSE=json.dumps({"crash_leg": leg}))
    r = subprocess.run([sys.executable, os.path.join(nb.ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=nb.ROOT)
    assert r.returncode == 6, (r.returncode, r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    out = json.loads(lines[0])
    assert out["extras_aborted"] == f"{leg} (fatal signal)" and out["value"] > 1e12 and out["roofline"]["frac"] > 0.3
    assert ("parity" in out) == (leg != "clock probe")      # legs that finished before the abort are on the line


def test_single_gpu_bench_skips_the_legs_its_budget_no_longer_holds():
    """--budget-s bounds the single-GPU run too: with a budget that covers the headline and little else, the optional legs
    that need more than what is left are LISTED (legs_skipped_for_budget) instead of started, the headline is unaffected, and
    the run reports success -- a slow box costs legs, never the line."""
    import json
    r = subprocess.run([sys.executable, os.path.join(nb.ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--budget-s", "14"],
                       capture_output=True, text=True, timeout=600, cwd=nb.ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["value"] > 1e12 and out["roofline"]["frac"] > 0.3 and "extras_aborted" not in out
    skipped = out["legs_skipped_for_budget"]
    assert "extra_configs C2/C3/N2/C1" in skipped and "extra_configs S2/S4/S8/C5S8" in skipped
    assert out.get("extra_configs", []) == []


@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_bench_shard_leg_times_every_ranks_step_and_stamps_it(ranks):
    """bench.py's S-legs (extra_configs S2 / S4 / S8 / C5S8) at a small size: all shards of one world in this process, every
    member's kernels under their own HIP event pairs (nb_hip_local_group_step with the timing knob), so the entry says what
    ONE rank's step costs for every rank, with a parity stamp against float64 and the stated gather estimate."""
    sys.path.insert(0, nb.ROOT)
    import bench
    n = 65536
    _, part, m = bench_universe(n)
    one = nb.SimPipeline(n, m)
    one.configure(graph=0)
    one.set_data(part)
    one.update(2, 0.01)
    t0 = time.perf_counter()
    one.update(5, 0.01)
    t1_ms = (time.perf_counter() - t0) / 5 * 1e3
    one.close()
    e = bench.shard_leg(nb, f"S{ranks}", part, m, ranks, 3, t1_ms=t1_ms)
    k = e["shard_kernel_ms_per_step"]
    assert e["ranks"] == ranks and 0 < k["min"] <= k["mean"] <= k["max"]                 # every member's kernels were timed
    assert e["single_gpu_ms_per_step"] == t1_ms and e["compute_scaling_efficiency"] > 0  # (how the times relate: test_gpu_zz_perf.py)
    p = e["parity"]
    assert p["worst_ratio"] <= 1.0 and p["integrator_bit_exact"] and p["static_fields_equal"], p
    plan = nb.shard_plan(n, m, 0, ranks)
    assert abs(e["gather_estimate_ms"] - bench.gather_estimate_ms(plan["mass_chunk"], ranks)) < 1e-12 and "NOT measured" in e["gather_estimate_source"]
    assert abs(e["predicted_steps_per_sec"] - 1e3 / (k["max"] + e["gather_estimate_ms"])) < 1e-6
