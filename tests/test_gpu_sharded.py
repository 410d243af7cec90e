"""The sharded (multi-GPU) pipeline as far as ONE GPU can run it (needs an MI355X: `pytest -m gpu`): all shards in one process (local
group), the RCCL path with one rank, several REAL processes on this GPU over the host-staged and the direct exchange, and bench.py's
multi-rank flow -- supervisor, transport chain, preflight, deadline and last-gasp lines (tolerances: tests/test_gpu_parity.py docstring).
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
# sharded pipeline on one GPU: local transport (all ranks in this process) and RCCL with one rank
# ---------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("P", [2, 3, 8])
@pytest.mark.parametrize("n", [4096, 333])
def test_local_shard_group_single_slice_is_bitwise_equal(golden, n, P):
    # w = 1: every receiver adds the sources in index order, pads add exact zeros -> same bits as one GPU
    part, m = ob.partition(golden(f"ic_{n}.bin"))
    want = run(part, m, 3, 0.01, w=1, k=1)
    g = nb.LocalShardGroup(n, m, P, w=1, k=1)
    g.set_data(part)
    g.step(3, 0.01)
    outs = [g.get_data(r) for r in range(P)]
    g.close()
    for o in outs:
        assert o.tobytes() == want.tobytes()


@pytest.mark.parametrize("overlap,split", [(0, 0), (1, 0), (0, 3)])
@pytest.mark.parametrize("P", [2, 8])
def test_local_shard_group_default_shape_within_tolerance(golden, P, overlap, split):
    part, m = ob.partition(golden("ic_4096.bin"))
    g = nb.LocalShardGroup(4096, m, P, overlap=overlap, split=split)
    g.set_data(part)
    g.step(1, 0.01)
    got = g.get_data(P - 1)
    g.step(9, 0.01)
    ten = g.get_data(0).astype(np.float64)
    g.close()
    check_one_step(got, part, m, 0.01)
    want = ob.step(part, m, 0.01, 10).astype(np.float64)
    assert np.linalg.norm(ten[:, 0:2] - want[:, 0:2]) / np.linalg.norm(want[:, 0:2]) <= 1e-6
    assert rel_displacement(ten, want, part) <= DISPLACEMENT_TOL


@pytest.mark.parametrize("overlap", [0, 1])
def test_local_shard_group_config4_shape(overlap):
    """BASELINE config 4/5 in miniature: 8 shards of a 65536-particle universe, spot-checked against float64."""
    n, P = 65536, 8
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic)
    part = w.particles()
    w.close()
    m = int((part[:, 6] > 0).sum())
    g = nb.LocalShardGroup(n, m, P, overlap=overlap)
    g.set_data(part)
    g.step(1, 0.01)
    got = g.get_data(3)
    g.close()
    idx = np.unique(np.random.default_rng(8).integers(0, n, 800)).astype(np.uint32)
    acc64, mag = ob.acc_f64_subset(part, m, idx)
    assert np.all(np.abs(got[idx, 4:6].astype(np.float64) - acc64) <= acc_bound(acc64, mag))
    v = part[:, 2:4] + got[:, 4:6] * np.float32(0.01)
    assert np.array_equal(got[:, 2:4], v) and np.array_equal(got[:, 0:2], part[:, 0:2] + v * np.float32(0.01))
    assert np.array_equal(got[:, 6:8], part[:, 6:8])


def test_local_shard_group_full_size_config4():
    """BASELINE config 4 at full size, all 8 shards on this one GPU: N = 2^20, two source passes per shard step."""
    n, P = 1 << 20, 8
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic)
    part = w.particles()
    w.close()
    m = int((part[:, 6] > 0).sum())
    g = nb.LocalShardGroup(n, m, P)
    g.set_data(part)
    g.step(1, 0.01)
    got = g.get_data(5)
    g.close()
    idx = np.unique(np.concatenate([[0, m - 1, m, n - 1], np.random.default_rng(3).integers(0, n, 400)])).astype(np.uint32)
    acc64, mag = ob.acc_f64_subset(part, m, idx)
    assert np.all(np.abs(got[idx, 4:6].astype(np.float64) - acc64) <= acc_bound(acc64, mag))
    v = part[:, 2:4] + got[:, 4:6] * np.float32(0.01)
    assert np.array_equal(got[:, 2:4], v) and np.array_equal(got[:, 0:2], part[:, 0:2] + v * np.float32(0.01))


def test_local_shard_group_full_size_config5_overlapped():
    """BASELINE config 5 at full size, all 8 shards on this one GPU: N = 2^22, own-slice kernel overlapped with
    the gather, then the remote-slice kernel (several source passes each)."""
    n, P = 1 << 22, 8
    ic = nb.make_galaxies(n, 2, seed=11037)
    w = nb.World(ic)
    part = w.particles()
    w.close()
    m = int((part[:, 6] > 0).sum())
    g = nb.LocalShardGroup(n, m, P, overlap=1)
    g.set_data(part)
    g.step(1, 0.01)
    got = g.get_data(2)
    g.close()
    idx = np.unique(np.concatenate([[0, m - 1, m, n - 1], np.random.default_rng(5).integers(0, n, 200)])).astype(np.uint32)
    acc64, mag = ob.acc_f64_subset(part, m, idx)
    assert np.all(np.abs(got[idx, 4:6].astype(np.float64) - acc64) <= acc_bound(acc64, mag))
    v = part[:, 2:4] + got[:, 4:6] * np.float32(0.01)
    assert np.array_equal(got[:, 2:4], v) and np.array_equal(got[:, 0:2], part[:, 0:2] + v * np.float32(0.01))
    assert np.array_equal(got[:, 6:8], part[:, 6:8])


@pytest.mark.parametrize("passes", [1, 2, 5])
def test_source_passes(golden, passes):
    # a step cut into `passes` launches over consecutive source sub-ranges, chained through acc[]
    part, m = ob.partition(golden("ic_4096.bin"))
    got = run(part, m, 1, 0.01, passes=passes)
    check_one_step(got, part, m, 0.01)
    assert run(part, m, 4, 0.01, passes=passes, graph=1).tobytes() == run(part, m, 4, 0.01, passes=passes, graph=0).tobytes()
    assert run(part, m, 1, 0.01, passes=passes, split=3).shape == part.shape


def test_local_shard_group_ragged(golden):
    part, m = synth(1000, 0.013, seed=4)      # 13 sources over 4 ranks: some ranks own no source
    g = nb.LocalShardGroup(1000, m, 4)
    g.set_data(part)
    g.step(1, 0.02)
    got = g.get_data(2)
    g.close()
    check_one_step(got, part, m, 0.02)


def test_rccl_path_with_one_rank_in_a_subprocess(golden, tmp_path):
    """NB_HIP_FORCE_SHARDED=1: dlopen librccl, ncclCommInitRank(1 rank), in-place all-gathers; plain, overlapped and
    hipGraph-captured chains."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
import nbody_amd as nb, oracle_binding as ob
ic = np.fromfile(os.path.join(%(root)r, "tests/golden/ic_1024.bin"), dtype=np.float32).reshape(-1, 8)
part, m = ob.partition(ic)
uid = nb.comm_unique_id()
L = nb.hip_lib()
outs = []
for overlap, sgraph in ((0, 0), (1, 0), (0, 1)):
    sim = nb.SimPipeline.__new__(nb.SimPipeline)
    import ctypes as C
    buf = (C.c_ubyte * 128).from_buffer_copy(nb.comm_unique_id())
    sim._h = L.CreateSimPipelineSharded(nb.WorldData(1024, m, 0.0), 0, 1, buf)
    sim.total_len, sim.mass_len, sim.rank, sim.nranks = 1024, m, 0, 1
    sim.configure(w=1, k=1, overlap=overlap, sharded_graph=sgraph)
    sim.set_data(part); sim.update(3, 0.01); sim.update(3, 0.01); outs.append(sim.get_data()); sim.close()
plain = nb.SimPipeline(1024, m); plain.configure(w=1, k=1); plain.set_data(part); plain.update(6, 0.01)
want = plain.get_data(); plain.close()
assert outs[0].tobytes() == want.tobytes(), "rccl 1-rank path differs"
assert outs[1].tobytes() == want.tobytes(), "rccl 1-rank overlap path differs"
assert outs[2].tobytes() == want.tobytes(), "rccl 1-rank captured-graph path differs"
print("RCCL-ONE-RANK-OK")
''' % {"root": nb.ROOT}
    env = dict(os.environ, NB_HIP_FORCE_SHARDED="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_one_rank_rccl_reports_its_communicator_and_gather_time(golden):
    """The evidence keys of a multi-GPU run, on the one rank a single-GPU box allows: ncclCommCount says 1, the probe
    all-gather was timed, and per-step kernel / gather intervals come back non-zero.  In a subprocess: the knob is an
    environment variable read at creation."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
import nbody_amd as nb, oracle_binding as ob
ic = np.fromfile(os.path.join(%(root)r, "tests/golden/ic_4096.bin"), dtype=np.float32).reshape(-1, 8)
part, m = ob.partition(ic)
sim = nb.SimPipeline(4096, m, rank=0, nranks=1, unique_id=nb.comm_unique_id())
info = sim.comm_info()
assert info["owns_comm"] and info["nranks"] == 1 and info["rank"] == 0 and info["rccl_version"] > 0, info
assert info["first_gather_ms"] > 0 and "rccl" in info["rccl_lib"], info
sim.set_data(part)
for overlap in (0, 1):
    sim.configure(overlap=overlap)
    sim.update(5, 0.01)
    steps, k_ms, c_ms = sim.step_breakdown()
    assert steps == 5 and k_ms > 0 and c_ms > 0, (overlap, steps, k_ms, c_ms)
    total, launches = sim.last_step_ms()
    assert total > 0 and k_ms <= total * 1.05
plain = nb.SimPipeline(4096, m)
assert plain.comm_info()["owns_comm"] is False and plain.step_breakdown()[0] == 0
plain.close(); sim.close()
print("RCCL-EVIDENCE-OK")
''' % {"root": nb.ROOT}
    env = dict(os.environ, NB_HIP_FORCE_SHARDED="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL-EVIDENCE-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_sharded_world_surface_with_one_rccl_rank(golden):
    """CreateWorldSharded (include/nbody.h extension) with the one rank this box has: the World's coherence protocol on
    top of the RCCL pipeline -- GPU steps, collective read-back, a CPU step on the gathered array, re-upload -- gives
    the ordinary World's state."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
import nbody_amd as nb
ic = np.fromfile(os.path.join(%(root)r, "tests/golden/ic_1024.bin"), dtype=np.float32).reshape(-1, 8)
def drive(w):
    out = []
    w.update_gpu(0.01, 2); out.append(w.particles())
    w.update_cpu(0.01, 1); w.update_gpu(0.01, 3); out.append(w.particles())
    w.update_gpu(0.005, 1); w.update_gpu(0.005, 1); out.append(w.particles())
    w.close()
    return out
plain = drive(nb.World(ic))
nb.hip_lib()
shard = drive(nb.World(ic, rank=0, nranks=1, unique_id=nb.comm_unique_id()))
for a, b in zip(plain, shard):
    d = a[:, 0:2].astype(np.float64) - b[:, 0:2]
    assert np.linalg.norm(d) / np.linalg.norm(a[:, 0:2].astype(np.float64)) <= 1e-7
    assert np.array_equal(a[:, 6:8], b[:, 6:8])
print("SHARDED-WORLD-OK")
''' % {"root": nb.ROOT}
    env = dict(os.environ, NB_HIP_FORCE_SHARDED="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SHARDED-WORLD-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("mode,rendezvous", [("plain", "socket"), ("sharded_graph", "socket"), ("plain", "gloo")])
def test_bench_under_torchrun_with_one_forced_sharded_rank(mode, rendezvous):
    """bench.py exactly as the driver launches it for N > 1 (python -m torch.distributed.run ... bench.py --gpus N),
    with the one rank this box has and NB_HIP_FORCE_SHARDED=1.  Default rendezvous (stdlib socket hub): torch is never
    imported, so the data path binds /opt/rocm's HIP runtime and librccl -- the stack the whole GPU suite runs on;
    `--rendezvous gloo` is round 3's route (torch first: its bundled runtime and RCCL).  Asserts the communicator
    evidence, non-zero gather time, the self-check against the plain single-GPU pipeline, and the extra_configs
    entries (plain + overlapped) at a second size."""
    import json
    env = dict(os.environ, NB_HIP_FORCE_SHARDED="1", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4",
               NB_HIP_SHARDED_GRAPH="1" if mode == "sharded_graph" else "0")
    port = {("plain", "socket"): "29731", ("sharded_graph", "socket"): "29732", ("plain", "gloo"): "29733"}[(mode, rendezvous)]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(nb.ROOT, "bench.py"), "--gpus", "1",
           "--steps", "4", "--warmup", "2", "--particles", "65536", "--extra-particles", "131072", "--rendezvous", rendezvous]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=nb.ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 1e11
    assert out["rccl_nranks"] == 1 and out["rccl"]["ranks_with_communicator"] == 1 and out["rccl"]["version"] > 0
    assert out["runtime"]["torch_imported_first"] is (rendezvous == "gloo") and out["runtime"]["hip_runtime_version"] > 0
    assert ("torch" in out["rccl"]["lib"]) == (rendezvous == "gloo"), out["rccl"]["lib"]     # which librccl the run bound
    check = out["self_check"]
    assert check["ranks_agree"] is True and check["static_fields_equal"] is True and check["steps"] == 6
    assert check["vs_single_gpu_rel_l2_pos"] <= 1e-7     # one rank: same sources, same order up to the launch shape
    if mode == "plain":
        assert out["comm_ms_per_step"]["max"] > 0 and out["kernel_ms_per_step"]["max"] > 0
    extra = out["extra_configs"]
    # overlapped step; config 5 x 2; the RCCL-free direct exchange on the headline workload (with its own self-check); and last
    # -- so that a stall there cannot cost the others -- the {kernel, ncclAllGather} x K chain captured as a hipGraph (north star)
    assert [(e["overlap"], e["sharded_graph"]) for e in extra] == [(1, 0), (0, 0), (1, 0), (0, 0), (0, 1)]
    assert extra[3]["transport"].startswith("direct") and extra[3]["self_check"]["ok"] is True
    assert extra[3]["cross_device_parity"].startswith("unpinned")      # one device here: nothing crossed xGMI
    assert all(e["value"] > 1e11 for e in extra) and "extras_aborted" not in out
    assert extra[4]["graph_stats"]["cached"] >= 1        # RCCL inside stream capture, instantiated and replayed
    assert all(e["comm_ms_per_step"]["max"] > 0 for e in extra if not e["sharded_graph"] and (e["overlap"] == 1 or mode == "plain"))


# ---------------------------------------------------------------------------------------------------------------
# several REAL processes through pipeline.hip's sharded host code on this one GPU (caller-supplied host transport)
# ---------------------------------------------------------------------------------------------------------------



# at most 3 ranks: the GPU boxes allow 6 processes on the card, and the pytest process and the torchrun launcher count too
@pytest.mark.parametrize("world,n,overlap", [(2, 1024, 0), (3, 333, 1), (2, 4096, 1), (3, 4096, 0)])
def test_sharded_pipeline_with_real_processes_on_one_gpu(golden, tmp_path, world, n, overlap):
    """pipeline.hip's sharded host code with `world` REAL processes (ranks > 0 in their own address space, collective
    Get included), all on this one GPU: the exchange goes through the caller-supplied host transport
    (CreateSimPipelineShardedWith) over gloo, because RCCL refuses two ranks on one device.  Everything but the
    ncclAllGather call itself is the RCCL path's code.  W = 1, gather in-stream: bit-equal to the single pipeline; auto shape: within
    the one-step tolerance chain (three steps, positions <= 1e-6 relative L2 of the single pipeline)."""
    worker = tmp_path / "worker.py"
    worker.write_text(_MULTI_PROC_WORKER)
    out = tmp_path / "out.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29800 + world), str(worker), nb.ROOT, str(out), str(n), str(overlap)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    got = np.load(out)
    part, m = ob.partition(golden(f"ic_{n}.bin"))
    single_w1 = run(part, m, 3, 0.01, w=1, k=1)
    if overlap == 0:
        assert got[0].tobytes() == single_w1.tobytes()
    else:
        # the overlapped step adds the rank's own slice first and the remote slices after it: another summation order
        assert rel_l2_pos(got[0], single_w1) <= 1e-6 and np.array_equal(got[0][:, 6:8], single_w1[:, 6:8])
    want = run(part, m, 3, 0.01)
    assert rel_l2_pos(got[1], want) <= 1e-6
    assert np.array_equal(got[1][:, 6:8], want[:, 6:8])


@pytest.mark.parametrize("transport", ["host", "direct"])
def test_bench_with_two_real_ranks_on_one_gpu(transport):
    """bench.py as the driver launches it for N = 2 -- two processes, barriers, reductions over the ranks, self-check
    against the single-GPU pipeline, extra_configs -- with both ranks on this one GPU over the host transport, and over
    the direct one (slices pushed device-to-device into IPC-mapped peers, one barrier per step over the socket hub)."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29741" if transport == "host" else "29742", os.path.join(nb.ROOT, "bench.py"), "--gpus", "2",
           "--transport", transport, "--steps", "4", "--warmup", "1", "--particles", "65536", "--extra-particles", "131072"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=nb.ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]      # stdout carries the JSON line only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 1e10 and out["rccl_nranks"] is None and out["transport"].startswith(transport)
    assert out["rccl"]["user_ranks"] == {"min": 0, "max": 1, "sum": 1} and out["rccl"]["ranks_with_communicator"] == 0
    check = out["self_check"]
    assert check["ranks_agree"] is True and check["static_fields_equal"] is True and check["steps"] == 5
    assert check["vs_single_gpu_rel_l2_pos"] <= 1e-6
    assert out["kernel_ms_per_step"]["min"] > 0 and out["comm_ms_per_step"]["max"] > 0
    extra = out["extra_configs"]
    assert [(e["overlap"], e["sharded_graph"]) for e in extra] == [(1, 0), (0, 0), (1, 0), (0, 1)]
    assert "skipped" in extra[3]              # a host callback cannot be captured into a hipGraph: RCCL transport only
    timed = [e for e in extra if "skipped" not in e]
    assert all(e["value"] > 1e10 and e["kernel_ms_per_step"]["max"] > 0 for e in timed)
    assert "extras_aborted" not in out
    # preflight: every rank's bring-up record, written before the headline -- PCI address, the
    # hipDeviceCanAccessPeer row, and (direct only: the host transport must not depend on IPC) one IPC open / close of the
    # next rank's exported word
    flights = out["preflight"]
    assert [f["rank"] for f in flights] == [0, 1] and all(f["transport"] == transport for f in flights)
    assert len({f["pci"] for f in flights}) == 1 and all(f["visible_devices"] >= 1 and f["can_access_peer"][0] == 1 for f in flights)
    if transport == "direct":
        assert all(f["ipc_export_rc"] == 0 and f["ipc_open_rc"] == 0 and f["ipc_open_peer"] == 1 - f["rank"] and f["ipc_open_ms"] > 0 for f in flights)
    else:
        assert all("ipc_open_rc" not in f for f in flights)
    trail = out["launch"]["attempts"][0]["preflight"]            # what the supervisors read off the workers' stderr, stage by stage
    assert {e["stage"] for e in trail} == ({"device", "ipc"} if transport == "direct" else {"device"})
    assert 0 < out["roofline"]["roofline_frac_from_wall"] < 1 and out["roofline"]["traffic_measured_in_this_run"] is False


def test_bench_line_survives_a_stuck_leg():
    """The first real multi-GPU run must not lose its headline to a stalled optional leg: the JSON dict is complete
    after the headline leg + self-check, every later leg runs under a host-side deadline, and on expiry rank 0 writes
    the line with what is in hand plus "extras_aborted" and every rank leaves with a fresh non-zero exit.  Rehearsed
    with two real ranks on this one GPU: during the 'overlap' leg the host transport's all-gather never returns."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", NB_BENCH_REHEARSE='{"stall_leg": "overlap"}')
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29743", os.path.join(nb.ROOT, "bench.py"), "--gpus", "2", "--transport", "host",
           "--steps", "4", "--warmup", "1", "--particles", "65536", "--extra-particles", "131072", "--leg-deadline-s", "10"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=nb.ROOT)
    assert r.returncode != 0, "a run whose leg stalled must not report success"
    assert time.time() - t0 < 300, "the leg's deadline, not the run's budget, must end the run"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["extras_aborted"] == "overlap"
    assert out["n_gpus"] == 2 and out["value"] > 1e10 and out["ms_per_step"] > 0      # the headline survived
    assert out["self_check"]["ranks_agree"] is True and out["self_check"]["vs_single_gpu_rel_l2_pos"] <= 1e-6
    assert out["extra_configs"] == []                                                   # no leg had finished yet
    assert "passed its deadline" in r.stderr


def test_bench_line_survives_a_leg_that_aborts():
    """... and not to an optional leg that dies by the library's own error convention (print + abort(), reference
    src/lib/util.h:17-29) either: rank 0's C-level handler writes the line prepared when the leg was armed."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", NB_BENCH_REHEARSE='{"crash_leg": "config5"}')
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29745", os.path.join(nb.ROOT, "bench.py"), "--gpus", "2", "--transport", "host",
           "--steps", "4", "--warmup", "1", "--particles", "65536", "--extra-particles", "131072"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=nb.ROOT)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["extras_aborted"] == "config5 (fatal signal)" and out["value"] > 1e10
    assert out["self_check"]["ranks_agree"] is True
    # the leg that had finished before the crash is on the line: the overlapped step
    assert [(e["overlap"], e["sharded_graph"]) for e in out["extra_configs"]] == [(1, 0)]
    assert "fatal signal 6" in r.stderr


def test_bench_auto_lands_on_the_host_transport_when_rccl_and_direct_are_refused():
    """--transport auto with two real ranks on this ONE GPU: RCCL refuses the duplicate device for real (both ranks abort in
    ncclCommInitRank), the direct attempt is made to fail right after its rendezvous (tests/bench_rehearsal.py), and the run
    lands on the transport nothing can refuse -- host-staged slices over the rank link -- with two transport_fallback
    entries, a verified headline and exit code 0.  Each failed attempt leaves its preflight trail."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(OMP_NUM_THREADS="4", NB_BENCH_REHEARSE='{"fail_transports": ["direct"]}')
    r = subprocess.run([sys.executable, os.path.join(nb.ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--particles", "65536", "--extra-particles", "131072", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=nb.ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert [a["transport"] for a in out["launch"]["attempts"]] == ["rccl", "direct", "host"]
    fb = out["transport_fallback"]
    assert [(f["from"], f["to"], f["kind"]) for f in fb] == [("rccl", "direct", "bring_up_failed"), ("direct", "host", "bring_up_failed")]
    assert out["transport"].startswith("host") and out["n_gpus"] == 2 and out["value"] > 1e10
    check = out["self_check"]
    assert check["ranks_agree"] is True and check["ok"] is True and check["vs_single_gpu_rel_l2_pos"] <= 1e-6
    # the RCCL attempt got as far as its IPC probe and announced ncclCommInitRank before it died there
    trail = out["launch"]["attempts"][0]["preflight"]
    assert any(e.get("stage") == "rccl" and "entering" in e for e in trail) and any(e.get("stage") == "ipc" and e.get("ipc_open_rc") == 0 for e in trail)
    assert out["launch"]["seconds"] < out["launch"]["budget_s"]


def test_bench_auto_survives_a_container_that_refuses_ipc_for_real():
    """The same chain with nothing rehearsed: HSA_ENABLE_IPC_MODE_LEGACY=1 selects the IPC mode this pool's host driver does
    not support, so hipIpcGetMemHandle fails with `invalid argument` -- a container refusing IPC.  Two ranks on this one
    GPU: RCCL refuses the duplicate device, the direct exchange aborts at its first IPC export, the host transport needs
    neither and delivers a verified headline.  The preflight of the failed attempts says WHY before anything aborted:
    ipc_export_rc != 0 with the runtime's own error string (profiles/r06_legacy_ipc_2ranks.json is this run, kept)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "NB_BENCH_REHEARSE")}
    env.update(OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="1")
    r = subprocess.run([sys.executable, os.path.join(nb.ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--particles", "65536", "--extra-particles", "131072", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=nb.ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    attempts = out["launch"]["attempts"]
    if len(attempts) == 2 and attempts[1]["child_rcs"] == [0, 0]:
        pytest.skip("this box's driver serves the legacy IPC mode: the direct exchange came up")
    assert r.returncode == 0, r.stderr[-3000:]
    assert [(a["transport"], a.get("kind")) for a in attempts] == [("rccl", "bring_up_failed"), ("direct", "bring_up_failed"), ("host", None)]
    for a in attempts[:2]:
        ipc = [e for e in a["preflight"] if e.get("stage") == "ipc" and "ipc_export_rc" in e]
        assert len(ipc) == 2 and all(e["ipc_export_rc"] != 0 and e["ipc_export_error"] and e["ipc_open_rc"] is None for e in ipc), a["preflight"]
        assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "1" for e in a["preflight"] if e.get("stage") == "device")
    assert "hipIpcGetMemHandle" in attempts[1]["stderr_tail"]          # the direct attempt died exactly where the preflight said it would
    assert all("ipc_export_rc" not in e for e in attempts[2]["preflight"])      # the host transport never asked
    assert out["transport"].startswith("host") and out["self_check"]["ok"] is True and out["value"] > 1e10
    assert [(f["from"], f["to"]) for f in out["transport_fallback"]] == [("rccl", "direct"), ("direct", "host")]


def test_host_transport_callback_that_raises_ends_the_process(golden, tmp_path):
    """A Python exception inside the caller-supplied all-gather must not escape into ctypes (it would be swallowed
    and the pipeline would step on stale peer slots): the thunk prints the traceback and leaves with exit code 5."""
    worker = tmp_path / "raises.py"
    worker.write_text(r'''
import os, sys, numpy as np
root = sys.argv[1]
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
import nbody_amd as nb, oracle_binding as ob
part, m = ob.partition(np.fromfile(os.path.join(root, "tests", "golden", "ic_333.bin"), dtype=np.float32).reshape(-1, 8))
def bad(rows, r, n):
    raise RuntimeError("transport fell over")
sim = nb.SimPipeline(333, m, rank=0, nranks=1, allgather=bad)
sim.set_data(part)
sim.update(1, 0.01)
print("NOT REACHED")
''')
    r = subprocess.run([sys.executable, str(worker), nb.ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 5, (r.returncode, r.stderr[-2000:])
    assert "transport fell over" in r.stderr and "all-gather raised" in r.stderr and "NOT REACHED" not in r.stdout
