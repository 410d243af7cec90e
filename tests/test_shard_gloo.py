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
    first = out["launch"]["attempts"][0]
    assert first["transport"] == "rccl" and first["child_rcs"] == [0] * world and "kind" not in first
    assert 0 < first["seconds"] < first["limit_s"] < out["launch"]["budget_s"] == 480.0       # the attempt's share of ONE budget
    # every rank's bring-up record: on the attempt (what the supervisor read off the workers' stderr) and on the line
    assert sorted(e["rank"] for e in first["preflight"]) == list(range(world))
    assert [e["rank"] for e in out["preflight"]] == list(range(world)) and all(e["transport"] == "rccl" for e in out["preflight"])
    assert "transport_fallback" not in out and "verification_failed" not in out and out["transport"].startswith("rccl")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "NB_BENCH_REHEARSE")}
    env["OMP_NUM_THREADS"] = "2"
    env.update(extra)
    return env


def _bench_cmd(launcher, world, port, *flags):
    root = os.path.dirname(HERE)
    tail = [os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--particles", "65536",
            "--extra-particles", "131072", "--dry-run", "--no-extras"] + list(flags)
    if launcher == "bare":
        return [sys.executable] + tail
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
            "--master-port", str(port)] + tail


@pytest.mark.parametrize("launcher,world", [("bare", 2), ("bare", 8), ("torchrun", 3)])
def test_bench_walks_the_transport_chain_in_fresh_processes(launcher, world):
    """--transport auto (the default) = rccl -> direct -> host: when a rank of an attempt leaves before the headline is in
    hand -- rehearsed through tests/bench_rehearsal.py: the last rank of the rccl AND of the direct attempt exits with 3 right
    after the rendezvous, what the library's watchdog does when a bring-up never completes -- the supervisor ends the
    attempt's other ranks by exact pid, starts a FRESH set of rank processes over the next transport and stamps the line with
    one transport_fallback entry per step.  Same behaviour whether bench.py spawned the ranks itself or
    torch.distributed.run did (then every rank process is a GPU-free supervisor of one worker)."""
    root = os.path.dirname(HERE)
    env = _clean_env(NB_BENCH_REHEARSE='{"fail_transports": ["rccl", "direct"]}')
    r = subprocess.run(_bench_cmd(launcher, world, 29677), env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = _one_json_line(r.stdout)
    a1, a2, a3 = out["launch"]["attempts"]
    assert [a["transport"] for a in (a1, a2, a3)] == ["rccl", "direct", "host"]
    for a in (a1, a2):
        assert a["kind"] == "bring_up_failed" and a["child_rcs"][world - 1] == 3 and a["why_not"]
    assert a3["child_rcs"] == [0] * world and "kind" not in a3
    assert a1["limit_s"] > a2["limit_s"] * 0.9 and a3["limit_s"] > a2["limit_s"]       # 1/2, 1/2 of the rest, then all of the rest
    fb = out["transport_fallback"]
    assert [(f["from"], f["to"], f["kind"]) for f in fb] == [("rccl", "direct", "bring_up_failed"), ("direct", "host", "bring_up_failed")]
    assert all(f["rc"][world - 1] == 3 and ("rc 3" in f["stderr_tail"] or "exit 3" in f["stderr_tail"]) for f in fb)
    assert out["transport"].startswith("host") and out["n_gpus"] == world and out["self_check"]["ranks_agree"] is True
    assert out["extra_configs"] == [] and "verification_failed" not in out      # --no-extras keeps the self-check, drops the legs


def test_bench_supervisor_reports_when_no_attempt_delivers():
    """An explicit --transport rccl has no fallback: the rehearsed failure ends the run with ONE line that says so (value
    null, the attempt's exit codes) and a non-zero exit code."""
    root = os.path.dirname(HERE)
    env = _clean_env(NB_BENCH_REHEARSE='{"fail_transports": ["rccl"]}')
    r = subprocess.run(_bench_cmd("bare", 2, 0, "--transport", "rccl"), env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode != 0
    out = _one_json_line(r.stdout)
    assert out["value"] is None and out["error"] and out["launch"]["attempts"][0]["child_rcs"][1] == 3
    assert len(out["launch"]["attempts"]) == 1 and "transport_fallback" not in out


@pytest.mark.parametrize("launcher,world", [("bare", 2), ("bare", 8), ("torchrun", 2)])
def test_bench_budget_ends_a_hung_run_with_a_line(launcher, world):
    """ONE total budget (--budget-s, default 480 s: under the 600 s a driver allows the command): every attempt's limit is
    carved from what is left.  Rehearsed: every rank of every attempt hangs right after the rendezvous (a first barrier that
    never completes).  The supervisor ends each attempt's ranks by exact pid when its share is used up, walks the whole chain,
    and the run still prints ONE line -- value null, the three attempts, why -- inside the budget, with a non-zero exit code
    (round 5 waited 900 s per attempt)."""
    import time
    root = os.path.dirname(HERE)
    env = _clean_env(NB_BENCH_REHEARSE='{"hang": true}')
    t0 = time.time()
    r = subprocess.run(_bench_cmd(launcher, world, 29679, "--budget-s", "30"), env=env, capture_output=True, text=True, timeout=600, cwd=root)
    took = time.time() - t0
    assert r.returncode != 0
    out = _one_json_line(r.stdout)
    assert out["value"] is None and "share of the budget" in out["error"]
    attempts = out["launch"]["attempts"]
    assert [a["transport"] for a in attempts] == ["rccl", "direct", "host"] and all(a["kind"] == "timed_out" for a in attempts)
    assert all(rc == -9 for a in attempts for rc in a["child_rcs"])                  # ended by the supervisor, by pid
    assert sum(a["seconds"] for a in attempts) <= 30.0 and out["launch"]["seconds"] <= 30.0
    if launcher == "bare":                   # torch.distributed.run's own start-up (a first `import torch` can take minutes) is not ours to budget
        assert took < 40.0
    assert [(f["from"], f["to"], f["kind"]) for f in out["transport_fallback"]] == [("rccl", "direct", "timed_out"), ("direct", "host", "timed_out")]


def test_bench_a_failed_self_check_is_not_a_bring_up_failure():
    """a wrong answer on the product transport must not end as exit 0 with a fallback comment.  Rehearsed: the rccl
    attempt delivers a complete line whose self-check says the ranks DISAGREE.  The supervisor still tries the next transport
    (the run yields a verified number), but the line carries "verification_failed" with the failed check, the fallback entry
    says kind = verification_failed -- not bring_up_failed -- and the exit code is non-zero."""
    root = os.path.dirname(HERE)
    env = _clean_env(NB_BENCH_REHEARSE='{"bad_self_check": ["rccl"]}')
    r = subprocess.run(_bench_cmd("bare", 2, 0), env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 5, (r.returncode, r.stderr[-2000:])
    out = _one_json_line(r.stdout)
    assert out["transport"].startswith("direct") and out["self_check"]["ranks_agree"] is True
    assert [v["transport"] for v in out["verification_failed"]] == ["rccl"] and out["verification_failed"][0]["self_check"]["ranks_agree"] is False
    assert [(f["from"], f["to"], f["kind"]) for f in out["transport_fallback"]] == [("rccl", "direct", "verification_failed")]
    assert out["launch"]["attempts"][0]["child_rcs"] == [0, 0]                       # nothing crashed: it was the CHECK that failed
    # ... and with no fallback to make, the unverified number is not printed as the result at all
    r = subprocess.run(_bench_cmd("bare", 2, 0, "--transport", "rccl"), env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 5
    out = _one_json_line(r.stdout)
    assert out["value"] is None and out["error"].startswith("self_check failed") and out["verification_failed"][0]["transport"] == "rccl"


def test_launch_budget_arithmetic_and_attempt_classes():
    """nbody_amd/launch.py without processes: how the budget is carved (1/2, 1/2 of the rest, all of the rest, minus the
    reserve kept for printing), and how a rank-0 line is classed: no line / a line without a value = bring-up failure; a
    complete line whose self-check failed = VERIFICATION failure (the line comes back with the reason); ok otherwise."""
    import json
    from nbody_amd import launch
    assert launch.TRANSPORT_CHAIN == ("rccl", "direct", "host")
    assert launch.carve(480.0, 3) == 237.5 and launch.carve(480.0 - 237.5, 2) == 118.75 and launch.carve(100.0, 1) == 95.0
    assert launch.carve(3.0, 2) == 0.0 and launch.carve(launch.PRINT_RESERVE_S + 2 * launch.MIN_ATTEMPT_S, 2) == launch.MIN_ATTEMPT_S
    ok = {"metric": "m", "value": 1.0, "self_check": {"ranks_agree": True, "ok": True}}
    assert launch.headline_of([json.dumps(ok)], 8, False) == (ok, None)
    null = {"metric": "m", "value": None, "error": "leg 'headline' passed its deadline"}
    assert launch.headline_of([json.dumps(null)], 8, False) == (None, "leg 'headline' passed its deadline")
    for bad in ({"ranks_agree": False}, {"ranks_agree": True, "ok": False}):
        line, why = launch.headline_of([json.dumps(dict(ok, self_check=bad))], 2, False)
        assert line is not None and why.startswith("self_check failed")
    line, why = launch.headline_of([json.dumps(dict(ok, self_check={"ranks_agree": False}))], 2, True)     # a dry run's check counts too
    assert line is not None and why.startswith("self_check failed")
    assert launch.headline_of([], 2, False) == (None, "rank 0 wrote no JSON line")


def test_bench_help_lists_no_rehearsal_flag_and_the_rehearsals_live_with_the_tests():
    """the timed tool carries no test-only flag or branch; the hooks are observers registered by
    tests/bench_rehearsal.py when NB_BENCH_REHEARSE is set, and an unknown rehearsal is refused loudly."""
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--budget-s" in r.stdout
    for word in ("--rehearse", "--stall", "--crash", "attempt-timeout"):
        assert word not in r.stdout
    src = open(os.path.join(root, "bench.py")).read()
    assert len(src.splitlines()) < 1200
    assert "os.abort" not in src and "rehearse_" not in src and "time.sleep(60" not in src and "time.sleep(3600" not in src
    sys.path.insert(0, root)
    import bench
    assert bench.OBSERVERS == []
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dry-run"], env=_clean_env(NB_BENCH_REHEARSE='{"typo": 1}'),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "unknown keys" in r.stderr


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
    import nbody_amd.benchlib as benchlib
    monkeypatch.setattr(benchlib, "kernel_sources_sha", lambda: "0" * 64)
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


def test_rank_link_hub_survives_silent_and_garbage_peers():
    """a same-uid stray that connects and stays silent gets HELLO_TIMEOUT_S, not the hub's whole timeout; one that
    sends bytes that are not a message (decode raises ValueError) or a frame that is too long is closed -- and in every case
    the hub keeps accepting, so the real rank still gets in."""
    import socket
    import struct
    import threading
    import time
    from nbody_amd import ranklink as rl
    name = f"nbody_test_strays_{os.getpid()}"
    result = {}

    def hub():
        t0 = time.time()
        link = rl.RankLink(0, 2, name=name, timeout_s=60)
        result["admitted_after"] = time.time() - t0
        result["gathered"] = link.allgather(b"hub")
        link.close()

    old = rl.HELLO_TIMEOUT_S
    rl.HELLO_TIMEOUT_S = 1.0
    try:
        t = threading.Thread(target=hub)
        t.start()
        path = os.path.join(rl.socket_dir(), name + ".sock")
        for _ in range(500):
            if os.path.exists(path):
                break
            time.sleep(0.01)
        silent = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        silent.connect(path)                       # says nothing: the hub must move on after HELLO_TIMEOUT_S
        garbage = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        garbage.connect(path)
        garbage.sendall(struct.pack("<Q", 3) + b"XYZ")                  # a frame that is not a message
        huge = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        huge.connect(path)
        huge.sendall(struct.pack("<Q", 1 << 40))                        # a frame longer than any message
        peer = rl.RankLink(1, 2, name=name, timeout_s=60)
        assert peer.allgather(b"peer") == [b"hub", b"peer"]
        peer.close()
        t.join(30)
    finally:
        rl.HELLO_TIMEOUT_S = old
    assert result["gathered"] == [b"hub", b"peer"] and result["admitted_after"] < 10.0
    for s_ in (silent, garbage, huge):
        s_.close()
    # a float list the receiver would refuse fails at the SENDER
    with pytest.raises(ValueError):
        rl.encode([0.0] * (rl._MAX_ITEMS + 1))
    # a hub nobody joins gives up at its own deadline with a message that says how many peers are missing
    with pytest.raises(TimeoutError, match="1 of 1 peers never reached the hub"):
        rl.RankLink(0, 2, name=name + "_alone", timeout_s=0.5)


def test_clock_sampler_bound_is_clamped_not_asserted():
    """bench.py asked the sampler for 3 x the headline leg + 0.5 s without a bound while the library asserted
    max_ms <= 20 000: a leg of ~6.5 s aborted the run.  Now the library clamps (a measurement aid never takes the process
    down over its own bound), the bound is one named constant in the header, and bench.py stays below it by itself."""
    import re
    root = os.path.dirname(HERE)
    import nbody_amd as nb
    header = open(os.path.join(root, "include", "nbody_hip.h")).read()
    assert float(re.search(r"#define NB_CLOCK_SAMPLER_MAX_MS ([0-9.]+)", header).group(1)) == nb.CLOCK_SAMPLER_MAX_MS == 20000.0
    src = open(os.path.join(root, "nbody_amd", "csrc", "clock_probe.hip")).read()
    begin = src[src.index('extern "C" int nb_hip_clock_sampler_begin'):]
    begin = begin[:begin.index("hipLaunchKernelGGL")]
    assert "NB_ASSERT(period_ms" not in begin and "max_ms = NB_CLOCK_SAMPLER_MAX_MS" in begin
    bench_src = open(os.path.join(root, "bench.py")).read()
    assert "min(nb.CLOCK_SAMPLER_MAX_MS, 3.0 * elapsed * 1e3 + 500.0)" in bench_src
    for elapsed_s in (0.1, 1.95, 6.5, 7.0, 400.0):       # 20 steps at 97 ms, ~70 steps, --particles 4194304 ...
        assert min(nb.CLOCK_SAMPLER_MAX_MS, 3.0 * elapsed_s * 1e3 + 500.0) <= 20000.0


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
    env = _clean_env(NB_BENCH_REHEARSE='{"hang": true}')
    sup = subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--particles", "65536", "--dry-run"],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root)
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
    sup = subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--particles", "65536", "--dry-run"],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root)
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
