"""bench_rehearsal -- failure rehearsals for bench.py, kept OUT of the timed tool (TEST INFRASTRUCTURE).

bench.py carries no rehearsal flag and no test-only branch: it announces what it is about to do through
`bench.notify(event, **info)` (an empty observer list in every real run).  When the environment variable NB_BENCH_REHEARSE
holds a JSON object, bench.main() imports this module and install() registers observers that make a chosen failure happen
at a chosen place -- in the worker processes only; the GPU-free supervisor is what is being tested and runs unmodified.

    {"hang": true}                      every rank sleeps for a minute right after the rendezvous (a collective that never
                                        completes): exercises the supervisor's budget and its clean-up
    {"fail_transports": ["rccl", ...]}  in an attempt over one of these transports the last rank leaves with exit code 3
                                        right after the rendezvous, like the library's watchdog when ncclCommInitRank never
                                        completes: exercises the rccl -> direct -> host chain
    {"bad_self_check": ["rccl", ...]}   the self-check of an attempt over one of these transports reports that the ranks
                                        disagree: a VERIFICATION failure (must not end as exit 0, whatever follows)
    {"stall_leg": "<leg>"}              host / direct transport: the all-gather callback never returns during this leg
    {"crash_leg": "<leg>"}              rank 0 abort()s inside this leg (N > 1: in the leg's all-gather; one GPU: when the leg starts)
"""
import json
import os
import sys
import time


def install(bench):
    spec = json.loads(os.environ["NB_BENCH_REHEARSE"])
    unknown = set(spec) - {"hang", "fail_transports", "bad_self_check", "stall_leg", "crash_leg"}
    if unknown:
        sys.exit(f"NB_BENCH_REHEARSE: unknown keys {sorted(unknown)}")

    def observer(event, info):
        if event == "rendezvous" and info.get("link") is not None:
            if spec.get("hang"):
                info["link"].barrier()
                time.sleep(60.0)
                os._exit(9)
            if info["transport"] in spec.get("fail_transports", ()) and info["world"] > 1:
                info["link"].barrier()
                if info["rank"] == info["world"] - 1:
                    print(f"[bench] rank {info['rank']}: rehearsing a {info['transport']} bring-up that never completes: exit 3",
                          file=sys.stderr, flush=True)
                    os._exit(3)
        elif event == "self_check" and info["transport"] in spec.get("bad_self_check", ()):
            info["check"]["ranks_agree"] = False
            info["check"]["rehearsed"] = "NB_BENCH_REHEARSE bad_self_check"
        elif event == "gather":
            if spec.get("stall_leg") == info["leg"]:
                time.sleep(3600.0)
            if spec.get("crash_leg") == info["leg"] and info["rank"] == 0:
                os.abort()
        elif event == "leg" and info.get("solo") and spec.get("crash_leg") == info["leg"]:
            os.abort()

    bench.OBSERVERS.append(observer)
