"""launch -- GPU-free supervision of the rank processes of a multi-GPU run, and the guards of a run's optional legs.

The reference's harness is one plain command that always prints its table (src/bench.c:41-74).  With several GPUs a run
can fail in ways a single process cannot -- RCCL that does not come up, an IPC open a container refuses, a collective that
never completes -- and the one rule that keeps such failures cheap is: NOTHING is retried or re-executed inside a process
that has touched the GPU.  So the process a user (or torch.distributed.run) starts never makes a GPU call:

  Supervisor   starts one fresh worker process per rank and attempt, with ONE total time budget (`--budget-s`, default 480 s:
               below the 600 s a driver allows the command) from which every attempt's limit, the rank link's timeout and
               the library's collective watchdog (NB_HIP_COMM_TIMEOUT_S) are carved; walks the transport chain
               rccl -> direct -> host (`--transport auto`): an attempt that does not deliver -- bring-up failure, time-out,
               or a self-check that FAILED -- is followed by the next transport in fresh processes; ends stragglers by exact
               pid; always prints one JSON line.  Exit code 0 only when the line's numbers are verified: a run that
               delivered after a failed self-check on an earlier transport still says so ("verification_failed") and
               leaves non-zero; a run whose later optional leg stalled (4) or aborted (6) keeps that code.
  LegGuard     host-side deadline around a leg that blocks inside a C call: rank 0 writes what is in hand, everybody exits 4.
  LastGasp     C-level handler (csrc/last_gasp.c): a fatal signal inside a leg still writes the line prepared beforehand.

Workers report their bring-up on stderr as `[preflight] {json}` lines (bench.py preflight): the supervisor keeps them per
attempt, so a first contact that fails says how far it got.
"""
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNNING = -1000   # status of a rank process that has not ended yet (exit codes and -signal numbers are > -1000)
PREFLIGHT_TAG = "[preflight] "
TRANSPORT_CHAIN = ("rccl", "direct", "host")   # --transport auto: the product path, then the two that need less of the platform
PRINT_RESERVE_S = 5.0      # kept back from the budget for writing the line and leaving
MIN_ATTEMPT_S = 5.0        # an attempt that would get less than this is not started


class LastGasp:
    """The legs of a run can die the hard way: the library's error convention is the reference's -- print and abort()
    (src/lib/util.h:17-29).  A fatal signal raised inside a C call never reaches a Python-level handler, so rank 0 registers
    a C one while legs run (csrc/last_gasp.c -> lib/libnb_lastgasp.so): on SIGABRT / SIGSEGV / SIGBUS / SIGFPE it write()s
    the line prepared when the current leg was armed and _exit(6)s.  Plain C, async-signal-safe calls on bytes copied
    beforehand: no GIL, no allocation, no Python in signal context."""

    def __init__(self, fd):
        self.fd = fd
        self._lib = C.CDLL(os.path.join(ROOT, "nbody_amd", "lib", "libnb_lastgasp.so"))
        self._lib.nb_last_gasp_set.argtypes = [C.c_int, C.c_char_p, C.c_ulong]
        self._lib.nb_last_gasp_set.restype = C.c_int

    def arm(self, line_bytes):
        if self._lib.nb_last_gasp_set(self.fd, line_bytes, len(line_bytes)) != 0:
            print("[bench] last-gasp line too long; keeping the previous one", file=sys.stderr)

    def disarm(self):
        self._lib.nb_last_gasp_disarm()


class LegGuard:
    """Host-side deadline around a leg.  Legs block inside C calls (ctypes releases the GIL), so a Python thread can watch
    the clock: on expiry rank 0 writes the JSON line with what is in hand (emit_partial(leg)), and every rank leaves with
    os._exit(4) -- a fresh exit (the process has touched the GPU; no re-exec, no retry).  The other ranks wait a moment
    first so that rank 0's line is out before the launcher reacts.  `hard_deadline` (time.monotonic() value or None) caps
    every leg: no leg is ever armed past the run's budget."""

    def __init__(self, rank, emit_partial, default_s, hard_deadline=None):
        self.rank, self.emit_partial, self.default_s, self.hard_deadline = rank, emit_partial, default_s, hard_deadline
        self.lock = threading.Lock()
        self.leg, self.until = None, None
        self.thread = threading.Thread(target=self._watch, daemon=True)
        self.thread.start()

    def arm(self, leg, seconds=None):
        until = time.monotonic() + (seconds if seconds else self.default_s)
        if self.hard_deadline is not None:
            until = min(until, self.hard_deadline)
        with self.lock:
            self.leg, self.until = leg, until

    def disarm(self):
        with self.lock:
            self.leg, self.until = None, None

    def _watch(self):
        while True:
            time.sleep(0.1)
            with self.lock:
                leg, until = self.leg, self.until
            if leg is None or time.monotonic() < until:
                continue
            print(f"[bench] rank {self.rank}: leg '{leg}' passed its deadline; writing what is in hand and exiting (4)",
                  file=sys.stderr, flush=True)
            if self.rank == 0:
                self.emit_partial(leg)
            else:
                time.sleep(3.0)
            os._exit(4)


def _die_with_parent():
    """In the child, between fork and exec: a rank never outlives the supervisor that started it, however that one ends
    (PR_SET_PDEATHSIG survives the exec; the signal handlers of supervise() cover the polite ways of being stopped)."""
    try:
        C.CDLL(None).prctl(1, 9)   # PR_SET_PDEATHSIG, SIGKILL
    except Exception:
        pass


class RankProcess:
    """One worker (the bench script, NB_BENCH_WORKER=1) for one rank of one attempt, started by a process that never touches
    the GPU.  Rank 0's stdout (the JSON line) is captured; every worker's stderr is forwarded as it comes, its tail kept, and
    its `[preflight] {json}` lines collected."""

    def __init__(self, script, argv, env, rank, capture_stdout):
        self.rank = rank
        self.lines, self.tail, self.preflight = [], [], []
        self.proc = subprocess.Popen([sys.executable, script] + argv, env=env, preexec_fn=_die_with_parent,
                                     stdout=subprocess.PIPE if capture_stdout else sys.stderr.fileno(), stderr=subprocess.PIPE)
        self.threads = [threading.Thread(target=self._pump_err, daemon=True)]
        if capture_stdout:
            self.threads.append(threading.Thread(target=self._pump_out, daemon=True))
        for t in self.threads:
            t.start()

    def _pump_out(self):
        for raw in self.proc.stdout:
            self.lines.append(raw.decode(errors="replace"))

    def _pump_err(self):
        for raw in self.proc.stderr:
            text = raw.decode(errors="replace")
            sys.stderr.write(text)
            sys.stderr.flush()
            if text.startswith(PREFLIGHT_TAG):
                try:
                    self.preflight.append(json.loads(text[len(PREFLIGHT_TAG):]))
                except ValueError:
                    pass
                continue
            self.tail.append(text)
            del self.tail[:-40]

    def status(self):
        rc = self.proc.poll()
        return RUNNING if rc is None else rc

    def end(self):
        """By exact pid: this Popen's own child, nothing matched by name."""
        if self.proc.poll() is None:
            self.proc.kill()

    def finish(self):
        self.proc.wait()
        for t in self.threads:
            t.join(5.0)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def headline_of(lines, world, dry_run):
    """(line dict or None, why not): the last JSON object rank 0 wrote, accepted when it is a complete headline -- metric and
    value present and, for a real multi-rank run, a self-check that passed.  A line whose self-check FAILED comes back
    together with the reason: it is a verification failure, not a bring-up failure."""
    for text in reversed(lines):
        text = text.strip()
        if not text.startswith("{"):
            continue
        try:
            line = json.loads(text)
        except ValueError:
            continue
        if "metric" not in line or line.get("value") is None:
            return None, line.get("error") or "rank 0 wrote a line without metric / value"
        if world > 1 and not dry_run:
            check = line.get("self_check")
            if not check:
                return None, "the line carries no self_check"
            if not check.get("ranks_agree") or check.get("ok") is False:
                return line, "self_check failed: " + json.dumps(check)
        elif world > 1 and line.get("self_check") and not line["self_check"].get("ranks_agree"):
            return line, "self_check failed: " + json.dumps(line["self_check"])
        return line, None
    return None, "rank 0 wrote no JSON line"


def carve(remaining_s, attempts_left):
    """What the next attempt may use of what is left of the budget: all of it when it is the last one, half when others may
    follow (so that rccl -> direct -> host get 1/2, 1/4, 1/4 of the total) -- never the reserve kept for printing."""
    usable = max(0.0, remaining_s - PRINT_RESERVE_S)
    return usable if attempts_left <= 1 else usable * 0.5


class Supervisor:
    """`bench.py --gpus N` (N > 1) started bare, or started once per rank by torch.distributed.run: THIS process never makes
    a GPU call.  Bare: it starts N fresh rank processes itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT in their
    environment) -- the shape of the reference harness, one plain command (src/bench.c:41-74).  Under torch.distributed.run:
    every rank process supervises ONE fresh worker and the supervisors keep each other informed over their own rank link."""

    GRACE_S = 20.0   # what the other ranks get once one has left with an error (rank 0 may be writing its line)

    def __init__(self, args, argv, script):
        self.args, self.script = args, script
        self.t0 = time.monotonic()
        self.deadline = self.t0 + float(args.budget_s)
        under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ
        if under_launcher:
            self.rank, self.world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
            self.local = [(self.rank, int(os.environ.get("LOCAL_RANK", self.rank)))]
            self.port = os.environ["MASTER_PORT"]
            self.mode = "torch.distributed.run: every rank process stays GPU-free and supervises one fresh worker"
        else:
            self.rank, self.world = 0, args.gpus
            self.local = [(r, r) for r in range(self.world)]
            self.port = str(free_port())
            self.mode = f"bare: bench.py started its {self.world} rank processes itself"
        self.run_id = os.environ.get("TORCHELASTIC_RUN_ID", "none")
        self.link = None
        if under_launcher and self.world > 1:
            from .ranklink import RankLink
            self.link = RankLink(self.rank, self.world, name=f"nbody_sup_{self.port}_{self.run_id}",
                                 timeout_s=max(10.0, float(args.budget_s)))
        # what the workers get: the same command line minus what the supervisor decides
        self.passthrough, skip = [], False
        for a in argv:
            if skip:
                skip = False
            elif a in ("--transport", "--budget-s"):
                skip = True
            elif not a.startswith(("--transport=", "--budget-s=")):
                self.passthrough.append(a)
        self.transports = list(TRANSPORT_CHAIN) if args.transport == "auto" else [args.transport]
        self.ranks, self.attempts = [], []

    def remaining(self):
        return self.deadline - time.monotonic()

    def everyone(self, values):
        """Status of every rank, indexed by rank (bare: they are all mine)."""
        if self.link is None:
            return list(values)
        return [int(x) for row in self.link.allgather([float(v) for v in values]) for x in row]

    def end_my_ranks(self):
        """Whatever ends this supervisor early -- a launcher's SIGTERM, an exception on the supervisors' link -- must not
        leave rank processes behind on the GPUs: end exactly the children this process started."""
        for p in self.ranks:
            p.end()

    def attempt(self, index, transport, limit_s):
        """One set of fresh rank processes over `transport`, for at most limit_s seconds: (line or None, why, kind, good).
        kind: "ok" | "verification_failed" (a line, but its self-check failed) | "timed_out" | "bring_up_failed"."""
        env = dict(os.environ, NB_BENCH_WORKER="1", WORLD_SIZE=str(self.world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(self.port),
                   NB_BENCH_ATTEMPT=str(index), TORCHELASTIC_RUN_ID=self.run_id,
                   NB_BENCH_DEADLINE_MONO=repr(time.monotonic() + limit_s))   # CLOCK_MONOTONIC is one clock for every process of the box
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the library's own bound on every wait for other ranks (default 180 s) must fit inside the attempt; with a fallback
        # behind it an attempt does not sit out more than 120 s (a cold box pages ~0.5 GB of librccl and its code objects in on
        # every rank before ncclCommInitRank returns: the bound must not mistake that for a hang).  A value the user exported is left alone.
        if "NB_HIP_COMM_TIMEOUT_S" not in os.environ:
            more = index + 1 < len(self.transports)
            env["NB_HIP_COMM_TIMEOUT_S"] = str(int(max(5.0, min(120.0 if more else 180.0, limit_s * (0.5 if more else 0.8)))))
        t0 = time.monotonic()
        self.ranks[:] = [RankProcess(self.script, self.passthrough + ["--transport", transport, "--budget-s", repr(limit_s)],
                                     dict(env, RANK=str(r), LOCAL_RANK=str(lr)), r, r == 0) for r, lr in self.local]
        first_failure, timed_out = None, False
        while True:
            seen = self.everyone([p.status() for p in self.ranks])
            if all(v != RUNNING for v in seen):
                break
            now = time.monotonic()
            if first_failure is None and any(v not in (RUNNING, 0) for v in seen):
                first_failure = now
            # a rank that left with an error takes the attempt with it: the others get a moment (rank 0 may be writing
            # its line; the library's own watchdogs may still fire), then go -- by exact pid.  So does an attempt that
            # has used up its share of the budget.
            if now - t0 > limit_s:
                timed_out = True
                self.end_my_ranks()
            elif first_failure is not None and now - first_failure > min(self.GRACE_S, max(1.0, limit_s / 8)):
                self.end_my_ranks()
            time.sleep(0.1)
        for p in self.ranks:
            p.finish()
        rcs = self.everyone([p.status() for p in self.ranks])
        timed_out = bool(max(self.everyone([1 if timed_out else 0] * len(self.ranks))))
        record = {"transport": transport, "child_rcs": rcs, "seconds": round(time.monotonic() - t0, 2), "limit_s": round(limit_s, 1)}
        # every rank's last words and bring-up trail, indexed by rank (under a launcher each supervisor holds one worker's)
        tails = ["".join(p.tail)[-900:] for p in self.ranks]
        flights = [p.preflight for p in self.ranks]
        if self.link is not None:
            tails = [t.decode(errors="replace") for t in self.link.allgather(tails[0].encode())]
            flights = [json.loads(b.decode()) for b in self.link.allgather(json.dumps(flights[0]).encode())]
        line, why, kind = None, None, "ok"
        if self.rank == 0:
            record["preflight"] = [e for f in flights for e in f]
            line, why = headline_of(self.ranks[0].lines, self.world, self.args.dry_run)
            if line is not None and why is not None:
                kind = "verification_failed"
            elif line is None:
                kind = "timed_out" if timed_out else "bring_up_failed"
                if timed_out:
                    why = f"attempt ended by the supervisor after its {limit_s:.0f} s share of the budget ({why})"
            if kind != "ok":
                # whose stderr explains it: a rank that left with something other than Python's generic 1, if there is one
                bad = sorted(range(self.world), key=lambda r: (rcs[r] == 0, rcs[r] == 1))[0]
                record.update({"kind": kind, "why_not": why, "stderr_tail": f"[rank {bad}, rc {rcs[bad]}] " + tails[bad]})
        good = 1 if kind == "ok" and line is not None else 0
        if self.link is not None:
            good = int(self.link.broadcast(good if self.rank == 0 else None, src=0))
        self.attempts.append(record)
        return line, why, kind, bool(good)

    def run(self):
        line, why, good = None, "no attempt ran", False
        unverified = []    # lines of attempts whose self-check failed: kept, never printed as the result
        for index, transport in enumerate(self.transports):
            limit = carve(self.remaining(), len(self.transports) - index)
            enough = limit >= MIN_ATTEMPT_S
            if self.link is not None:
                enough = bool(int(self.link.broadcast(int(enough) if self.rank == 0 else None, src=0)))
            if not enough:
                why = f"budget of {self.args.budget_s:g} s used up before the {transport} attempt ({why})"
                self.attempts.append({"transport": transport, "skipped": "no budget left"})
                break
            line, why, kind, good = self.attempt(index, transport, limit)
            if kind == "verification_failed" and self.rank == 0:
                unverified.append({"transport": transport, "self_check": line.get("self_check"), "value": line.get("value")})
            if good:
                break
        if self.link is not None:
            self.link.barrier()
            self.link.close()
        ran = [a for a in self.attempts if "child_rcs" in a]
        last = [rc for rc in (ran[-1]["child_rcs"] if ran else [1]) if rc not in (0, RUNNING)]
        code = (min(abs(last[0]), 255) or 1) if last else 0      # the worst the final attempt's ranks left with; 0 only when all did
        if self.rank != 0:
            return code if good else (code or 1)
        launch = {"mode": self.mode, "budget_s": float(self.args.budget_s), "seconds": round(time.monotonic() - self.t0, 2),
                  "attempts": self.attempts}
        fallbacks = [{"from": a["transport"], "to": b["transport"], "kind": a.get("kind"), "rc": a.get("child_rcs"), "why": a.get("why_not"),
                      "stderr_tail": a.get("stderr_tail", "")[-600:]} for a, b in zip(self.attempts, self.attempts[1:]) if "child_rcs" in b]
        if good:
            line["launch"] = launch
            if fallbacks:
                line["transport_fallback"] = fallbacks
            if unverified:
                # a wrong answer on an earlier transport is not repaired by a right one on a later transport: say so, and do
                # not report success (a verification failure is not a bring-up failure)
                line["verification_failed"] = unverified
                code = code or 5
            print(json.dumps(line), flush=True)
            return code
        # no complete, verified headline from any attempt: still one line, saying so
        partial = {"metric": "particle-pair interactions/sec at N=2^20", "value": None, "unit": "interactions/s", "n_gpus": self.world,
                   "error": why, "launch": launch}
        if fallbacks:
            partial["transport_fallback"] = fallbacks
        if unverified:
            partial["verification_failed"] = unverified
        print(json.dumps(partial), flush=True)
        return code or (5 if unverified else 1)


def supervise(args, argv, script):
    import signal
    sup = Supervisor(args, argv, script)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, lambda n, f: (sup.end_my_ranks(), os._exit(128 + n)))
    try:
        return sup.run()
    finally:
        sup.end_my_ranks()
