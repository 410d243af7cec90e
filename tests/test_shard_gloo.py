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


@pytest.mark.parametrize("world,rendezvous", [(2, "socket"), (3, "socket"), (8, "socket"), (2, "gloo")])
def test_bench_multi_rank_control_flow_dry_run(world, rendezvous):
    """bench.py under torch.distributed.run, world_size > 1, no GPU: rendezvous (the stdlib socket hub by default, gloo on
    request), RCCL-id broadcast, barriers, max-over-ranks reduction and the one JSON line on rank 0 (the sharded device
    work is skipped)."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29610 + world + (10 if rendezvous == "gloo" else 0)),
           os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--particles", "65536",
           "--extra-particles", "131072", "--dry-run", "--rendezvous", rendezvous]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "strong"
    assert out["unit"] == "interactions/s" and out["dtype"] == "f32" and "roofline" in out and "cpu_baseline" not in out
    assert f"N/{world}" in out["config"]["parallelism"]
    # what the first real multi-GPU run must carry: the communicator's own rank count, the
    # per-step all-gather and kernel times over the ranks, the cross-rank self-check, and config 5 from the same command
    assert "rccl_nranks" in out and set(out["rccl"]) >= {"nranks_reported", "user_ranks", "devices", "version", "lib",
                                                         "ranks_with_communicator", "first_gather_ms_rank0"}
    for key in ("comm_ms_per_step", "kernel_ms_per_step"):
        assert set(out[key]) == {"min", "max"}
    assert out["self_check"]["ranks_agree"] is True          # every rank generated the same initial conditions
    extra = out["extra_configs"]
    # overlapped step, config 5 plain and overlapped, the direct exchange on the headline workload, and LAST -- a stall there
    # must not cost the others -- the chain captured as a hipGraph (north star; RCCL inside stream capture)
    assert [(e["overlap"], e["sharded_graph"], "N=65536" in e["workload"]) for e in extra] == \
        [(1, 0, True), (0, 0, False), (1, 0, False), (0, 0, True), (0, 1, True)]
    assert all("N=131072" in e["workload"] for e in extra[1:3])
    assert extra[3]["transport"].startswith("direct")
    for e in extra:
        assert set(e) >= {"workload", "overlap", "sharded_graph", "steps", "ms_per_step", "steps_per_sec", "value", "unit",
                          "kernel_ms_per_step", "comm_ms_per_step"}
    assert "extras_aborted" not in out
    assert out["roofline"]["traffic"] is None and "traffic_note" in out["roofline"]


def _one_json_line(stdout):
    import json
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 8])
def test_bench_started_bare_spawns_its_own_ranks(world):
    """`python bench.py --gpus N` with no launcher around it (the shape of the driver's N = 1 command, and of the reference
    harness: one plain process, src/bench.c:41-74): the process stays GPU-free, starts N fresh rank processes itself, relays
    rank 0's one line and says on it how the ranks came to be."""
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
                        "--particles", "65536", "--extra-particles", "131072", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = _one_json_line(r.stdout)
    assert out["n_gpus"] == world and out["self_check"]["ranks_agree"] is True
    assert out["launch"]["mode"].startswith("bare") and len(out["launch"]["attempts"]) == 1
    assert out["launch"]["attempts"][0] == {"transport": "rccl", "child_rcs": [0] * world,
                                            "seconds": out["launch"]["attempts"][0]["seconds"]}
    assert "transport_fallback" not in out and out["transport"].startswith("rccl")


@pytest.mark.parametrize("launcher", ["bare", "torchrun"])
def test_bench_falls_back_to_the_direct_exchange_in_fresh_processes(launcher):
    """--transport auto (the default): when a rank of the RCCL attempt leaves before the headline is in hand -- rehearsed: the
    last rank exits with 3 right after the rendezvous, what the library's watchdog does when ncclCommInitRank never
    completes -- the supervisor ends the attempt's other ranks by exact pid, starts a FRESH set of rank processes over the
    direct exchange and stamps the line.  Same behaviour whether bench.py spawned the ranks itself or
    torch.distributed.run did (then every rank process is a GPU-free supervisor of one worker)."""
    root = os.path.dirname(HERE)
    world = 3
    tail = [os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--particles", "65536",
            "--extra-particles", "131072", "--dry-run", "--no-extras", "--rehearse-rccl-failure"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "2"
    if launcher == "bare":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", "29677"] + tail
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = _one_json_line(r.stdout)
    first, second = out["launch"]["attempts"]
    assert first["transport"] == "rccl" and first["child_rcs"][world - 1] == 3 and any(rc != 0 for rc in first["child_rcs"])
    assert second["transport"] == "direct" and second["child_rcs"] == [0] * world
    fb = out["transport_fallback"]
    assert fb["from"] == "rccl" and fb["to"] == "direct" and fb["rc"] == first["child_rcs"] and fb["why"]
    assert "rc 3" in fb["stderr_tail"] or "exit 3" in fb["stderr_tail"]
    assert out["transport"].startswith("direct") and out["n_gpus"] == world and out["self_check"]["ranks_agree"] is True
    assert out["extra_configs"] == []       # --no-extras keeps the self-check and drops the optional legs


def test_bench_supervisor_reports_when_no_attempt_delivers():
    """An explicit --transport rccl has no fallback: the rehearsed failure ends the run with ONE line that says so (value
    null, the attempt's exit codes) and a non-zero exit code."""
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--particles", "65536",
                        "--dry-run", "--no-extras", "--rehearse-rccl-failure", "--transport", "rccl"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode != 0
    out = _one_json_line(r.stdout)
    assert out["value"] is None and out["error"] and out["launch"]["attempts"][0]["child_rcs"][1] == 3
    assert len(out["launch"]["attempts"]) == 1 and "transport_fallback" not in out


def test_bench_traffic_figure_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC profile and must go null (with a note) once the kernel sources no
    longer hash to what was profiled."""
    import json
    root = os.path.dirname(HERE)
    sys.path.insert(0, root)
    import bench
    rec = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    assert set(rec) >= {"hbm_bytes_per_launch", "kernel_sources_sha256", "n", "source", "launch"}
    value, note = bench.pmc_traffic(1 << 20)
    if rec["kernel_sources_sha256"] == bench.kernel_sources_sha():
        assert value == rec["hbm_bytes_per_launch"] and "from profiles/" in note
    else:
        assert value is None and note.startswith("stale")
    # a run that launched another shape, or cut the step into another number of source passes, gets no figure either
    launch = dict(rec["launch"])
    passes = launch.pop("passes")
    if rec["kernel_sources_sha256"] == bench.kernel_sources_sha():
        assert bench.pmc_traffic(1 << 20, launch, passes)[0] == rec["hbm_bytes_per_launch"]
    assert bench.pmc_traffic(1 << 20, dict(launch, split=launch["split"] + 1), passes)[1].startswith("stale: this run launched")
    assert bench.pmc_traffic(1 << 20, launch, passes + 1)[0] is None
    monkeypatch.setattr(bench, "kernel_sources_sha", lambda: "0" * 64)
    value, note = bench.pmc_traffic(1 << 20)
    assert value is None and note.startswith("stale")
    assert bench.pmc_traffic(12345)[0] is None
    # N * (12 + 12 + 8 + 8) + 12 M read, N * 32 written, over two passes
    assert bench.algorithmic_bytes_per_launch(1 << 20, 523884, 2) == ((1 << 20) * 72 + 523884 * 12) / 2


def test_bench_last_gasp_line_on_a_fatal_signal():
    """bench.py's C-level handler (LastGasp): a process that dies by abort() inside a C call still writes the line
    that was prepared beforehand, exactly once, and leaves with exit code 6 (no GPU needed)."""
    import textwrap
    root = os.path.dirname(HERE)
    code = textwrap.dedent(f'''
        import ctypes, os, sys
        sys.path.insert(0, {root!r})
        import bench
        g = bench.LastGasp(1)
        g.arm(b'{{"value": 1, "extras_aborted": "overlap (fatal signal)"}}\\n')
        g.arm(b'{{"value": 2, "extras_aborted": "config5 (fatal signal)"}}\\n')   # the later leg replaces the earlier line
        ctypes.CDLL(None).abort()
    ''')
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 6, (r.returncode, r.stderr[-500:])
    assert r.stdout == '{"value": 2, "extras_aborted": "config5 (fatal signal)"}\n'
    assert "fatal signal 6" in r.stderr
    # disarmed: the default disposition is not restored, but nothing is written twice and the exit code still says "died"
    code2 = code.replace("ctypes.CDLL(None).abort()", "g.disarm(); ctypes.CDLL(None).abort()")
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=120)
    assert r.returncode == 6 and r.stdout == ""


def test_bench_leg_guard_deadline_writes_the_line_and_exits():
    """bench.py's LegGuard without a GPU: a leg that outlives its deadline makes rank 0 write what is in hand with
    "extras_aborted" and leave with exit code 4; other ranks leave with 4 and write nothing; a disarmed guard stays quiet."""
    import json
    import textwrap
    root = os.path.dirname(HERE)
    code = textwrap.dedent(f'''
        import json, os, sys, time
        sys.path.insert(0, {root!r})
        import bench
        rank = int(sys.argv[1])
        out = {{"value": 123.0, "self_check": {{"ranks_agree": True}}, "extra_configs": []}}
        def emit(leg):
            os.write(1, (json.dumps(dict(out, extras_aborted=leg)) + "\\n").encode())
        g = bench.LegGuard(rank, emit, default_s=0.5)
        g.arm("fast")
        g.disarm()
        time.sleep(1.0)                 # disarmed: nothing may fire
        g.arm("overlap")
        time.sleep(30)                  # "stuck in a C call"
        print("NOT REACHED")
    ''')
    t0 = __import__("time").time()
    r0 = subprocess.run([sys.executable, "-c", code, "0"], capture_output=True, text=True, timeout=120)
    assert r0.returncode == 4 and __import__("time").time() - t0 < 20
    lines = r0.stdout.splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"value": 123.0, "self_check": {"ranks_agree": True},
                                                         "extra_configs": [], "extras_aborted": "overlap"}
    assert "passed its deadline" in r0.stderr
    r1 = subprocess.run([sys.executable, "-c", code, "1"], capture_output=True, text=True, timeout=120)
    assert r1.returncode == 4 and r1.stdout == ""


@pytest.mark.parametrize("world", [1, 2, 4])
def test_rank_link_collectives_with_real_processes(world, tmp_path):
    """nbody_amd/ranklink.py: `world` processes meet over the Unix socket in the user's 0700 directory, then all-gather,
    broadcast, reduce and barrier; no torch and no pickle in the workers (asserted: importing torch is what the link exists to
    avoid, and nothing received is ever unpickled)."""
    root = os.path.dirname(HERE)
    code = f"""
import sys
sys.path.insert(0, {root!r})
from nbody_amd.ranklink import RankLink
rank, world = int(sys.argv[1]), int(sys.argv[2])
link = RankLink(rank, world, name="nbody_test_{os.getpid()}_" + sys.argv[2], timeout_s=60)
assert link.allgather(rank) == list(range(world))
assert link.allgather([float(rank), 0.5]) == [[float(q), 0.5] for q in range(world)]
assert link.broadcast(b"x" * 128 if rank == 0 else None) == b"x" * 128
assert link.reduce([rank, -rank, 1.0], "max") == [world - 1.0, 0.0, 1.0]
assert link.reduce([rank, -rank, 1.0], "min") == [0.0, -(world - 1.0), 1.0]
assert link.reduce([rank, 2.0], "sum") == [world * (world - 1) / 2.0, 2.0 * world]
big = bytes([rank]) * (3 << 20)
rows = link.allgather(big)
assert [len(r) for r in rows] == [3 << 20] * world and all(rows[q][:1] == bytes([q]) for q in range(world))
for _ in range(100):
    link.barrier()
link.close()
assert "torch" not in sys.modules
import inspect, nbody_amd.ranklink as rl
assert "pickle" not in inspect.getsource(rl).replace("unpickled", "")
print("ok", rank)
"""
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(world)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert sorted(o[0].strip() for o in outs) == [f"ok {r}" for r in range(world)]


def test_rank_link_a_dying_rank_ends_the_others_quickly():
    """A rank that dies mid-run must not leave the others blocked until the 900 s socket timeout: rank 1 exits before the
    second collective, the hub (rank 0) sees its socket close and raises, rank 2 sees the hub's close and raises."""
    import time
    root = os.path.dirname(HERE)
    code = f"""
import os, sys
sys.path.insert(0, {root!r})
from nbody_amd.ranklink import RankLink
rank = int(sys.argv[1])
link = RankLink(rank, 3, name="nbody_test_die_{os.getpid()}", timeout_s=120)
link.barrier()
if rank == 1:
    os._exit(7)
link.barrier()
link.barrier()
print("NOT REACHED")
"""
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(3)]
    outs = [p.communicate(timeout=100) for p in procs]
    assert time.time() - t0 < 30
    assert procs[1].returncode == 7
    assert procs[0].returncode != 0 and procs[2].returncode != 0
    assert all("NOT REACHED" not in o[0] for o in outs)
    assert "ConnectionError" in outs[0][1] or "Connection" in outs[0][1] or "Broken" in outs[0][1]


def test_rank_link_wire_format_is_closed():
    """The link's wire format carries None, bytes, int64, float64 lists and one level of list -- and nothing else: what is
    not in the format is refused at encode time, and bytes that are not a valid message raise instead of being interpreted
    (no pickle on a local socket)."""
    from nbody_amd import ranklink as rl
    for obj in (None, b"", b"abc" * 1000, 0, -5, 2 ** 62, [1.0, -2.5], [], [None, b"x", 3, [0.25]], [[1.0], [2.0, 3.0]]):
        data = rl.encode(obj)
        back, end = rl.decode(data)
        assert end == len(data) and back == (list(obj) if isinstance(obj, tuple) else obj)
    for bad in ("text", 1.5, {"a": 1}, True, [[[1.0]]], [object()]):
        with pytest.raises((TypeError, ValueError)):
            rl.encode(bad)
    import pickle
    for junk in (b"", b"X", b"B\xff\xff\xff\xff\xff\xff\xff\x7f", b"F\xff\xff\xff\xff", b"L\x01\x00\x00\x00L\x00\x00\x00\x00",
                 b"I\x00", pickle.dumps({"a": 1})):
        with pytest.raises(ValueError):
            rl.decode(junk)


def test_rank_link_socket_lives_in_a_private_directory_and_refuses_strangers(tmp_path):
    """The hub's socket is a file in a 0700 directory of this user, and a connection that does not answer the hub's nonce
    with the run's token is dropped without taking a rank slot; the real rank still gets in afterwards."""
    import socket
    import stat
    import threading
    import time
    from nbody_amd import ranklink as rl
    d = rl.socket_dir()
    st = os.lstat(d)
    assert stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0
    name = f"nbody_test_stranger_{os.getpid()}"
    result = {}

    def hub():
        link = rl.RankLink(0, 2, name=name, timeout_s=60)
        result["gathered"] = link.allgather(b"hub")
        link.close()

    t = threading.Thread(target=hub)
    t.start()
    path = os.path.join(d, name + ".sock")
    for _ in range(500):
        if os.path.exists(path):
            break
        time.sleep(0.01)
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.connect(path)
    nonce = rl.RankLink._recv(s)
    assert isinstance(nonce, bytes) and len(nonce) == 16
    rl.RankLink._send(s, [1, b"\0" * 32])      # claims rank 1 with a wrong token
    s.settimeout(10)
    assert s.recv(1) == b""                     # dropped
    s.close()
    peer = rl.RankLink(1, 2, name=name, timeout_s=60)
    assert peer.allgather(b"peer") == [b"hub", b"peer"]
    peer.close()
    t.join(30)
    assert result["gathered"] == [b"hub", b"peer"]
    assert not os.path.exists(path)             # the name is free again once everybody is connected


def test_bench_pure_helpers_for_the_round5_line():
    """The arithmetic bench.py puts on the line, checked without a GPU: cycles per wave-interaction from kernel seconds and
    the held clock, the fraction of the instruction mix's 26-cycle floor, the gather estimate with its stated parts, the
    parsing of nb_hip_device_info, and which rank-0 output the supervisor accepts as a complete headline."""
    import json
    root = os.path.dirname(HERE)
    sys.path.insert(0, root)
    import bench
    info = "AMD Instinct MI355X gfx950:sramecc+:xnack- 256 2400 pci=0000:f1:00"
    assert bench.device_cus(info) == 256 and bench.device_cus("AMD-GPU gfx950 304 2100") == 304 and bench.device_cus("?") == 256
    # one launch of the headline: 2.7465e11 interactions in 47.88 ms at 2.354 GHz on 1024 SIMDs -> 26.9 cycles
    sampled = {"clock_ghz": 2.35, "per_xcd_ghz": [2.373, 2.340, 2.371, 2.338, 2.377, 2.331, 2.378, 2.322]}
    f = bench.held_clock_fields({"clock_ghz": 2.27}, sampled, 47.88e-3, 2.7465e11, info, 80.3)
    assert abs(f["held_clock_ghz"] - 2.35375) < 1e-9 and abs(f["held_clock_ghz_slowest_xcd"] - 2.322) < 1e-12
    assert abs(f["cycles_per_wave_interaction"] - 26.89) < 0.02 and abs(f["frac_of_mix_ceiling"] - 26.0 / f["cycles_per_wave_interaction"]) < 1e-12
    assert abs(f["cycles_per_wave_interaction_slowest_xcd"] - 26.53) < 0.02
    assert abs(f["mix_ceiling_frac_at_nominal_clock"] - 14.0 / 26.0) < 1e-12 and abs(f["frac_at_held_clock"] - 80.3 / (157.3 * 2.35375 / 2.4)) < 1e-12
    # the probe alone (no sampler leg): its clock is used, and labelled as reading low
    g = bench.held_clock_fields({"clock_ghz": 2.27}, None, 47.88e-3, 2.7465e11, info, 80.3)
    assert g["held_clock_ghz"] == 2.27 and "LOW" in g["held_clock_source"]
    assert bench.held_clock_fields(None, None, 47.88e-3, 2.7465e11, info, 80.3)["held_clock_ghz"] is None
    # gather estimate: the slice over its own link + the assumed fixed latency, nothing for one rank
    assert bench.gather_estimate_ms(65536, 1) == 0.0
    est = bench.gather_estimate_ms(65536, 8)
    assert abs(est - (65536 * 8 / 153e9 * 1e3 + bench.GATHER_LATENCY_ASSUMED_MS)) < 1e-12 and 0.05 < est < 0.06
    # what counts as a headline: metric + value, and for real multi-rank runs a self-check that passed
    ok = json.dumps({"metric": "m", "value": 1.0, "self_check": {"ranks_agree": True, "ok": True}})
    assert bench._headline_of(["noise\n", ok + "\n"], 8, False)[1] is None
    assert bench._headline_of([ok], 1, False)[1] is None
    assert "no self_check" in bench._headline_of([json.dumps({"metric": "m", "value": 1.0})], 2, False)[1]
    assert bench._headline_of([json.dumps({"metric": "m", "value": 1.0})], 2, True)[1] is None           # dry run
    bad = json.dumps({"metric": "m", "value": 1.0, "self_check": {"ranks_agree": False}})
    assert bench._headline_of([bad], 2, False)[1].startswith("self_check failed")
    assert bench._headline_of(["not json\n"], 2, False) == (None, "rank 0 wrote no JSON line")
    share = bench.host_cpu_share()
    assert share["threads"] >= 1 and share["threads_from"] and share["os_cpu_count"] >= 1


def test_bench_supervisor_takes_its_ranks_with_it_when_told_to_stop():
    """A supervisor that is ended from outside (a launcher's SIGTERM when another rank failed, Ctrl-C) must not leave rank
    processes behind on the GPUs: it ends exactly the children it started, then leaves."""
    import signal
    import time
    import psutil
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    sup = subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--particles", "65536", "--dry-run",
                            "--rehearse-hang"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root)
    kids = []
    for _ in range(200):
        kids = psutil.Process(sup.pid).children(recursive=True)
        if len(kids) == 3:
            break
        time.sleep(0.05)
    assert len(kids) == 3, kids
    time.sleep(1.0)
    sup.send_signal(signal.SIGTERM)
    assert sup.wait(timeout=30) == 128 + signal.SIGTERM
    gone, alive = psutil.wait_procs(kids, timeout=10)
    assert not alive, alive
    # ... and not even a SIGKILL of the supervisor leaves them behind (PR_SET_PDEATHSIG in every rank process)
    sup = subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--particles", "65536", "--dry-run",
                            "--rehearse-hang"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root)
    for _ in range(200):
        kids = psutil.Process(sup.pid).children(recursive=True)
        if len(kids) == 2:
            break
        time.sleep(0.05)
    assert len(kids) == 2, kids
    sup.kill()
    sup.wait(timeout=30)
    gone, alive = psutil.wait_procs(kids, timeout=10)
    assert not alive, alive
