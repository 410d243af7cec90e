#!/usr/bin/env python3
"""Condense gpurun_out/prof_mid/ (tools/profile_mid_n.sh) into profiles/<tag>_mid_n_pmc.txt: per (kernel, grid) the
mean duration and SQ counters, plus what they say about the latency-bound regime (waves per SIMD, VALU issue share of
the busy time, share of wave time spent waiting)."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PROF = os.path.join(ROOT, "gpurun_out", "prof_mid")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
suffix = sys.argv[2] if len(sys.argv) > 2 else "mid_n_pmc"   # output: profiles/<tag>_<suffix>.txt
sizes = sys.argv[3] if len(sys.argv) > 3 else "--n 10000 --n 20000 --n 50000"
agg = collections.defaultdict(lambda: collections.defaultdict(list))


def newest(pattern):
    best = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


for f in newest(os.path.join(PROF, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.split("(anonymous namespace)::")[-1].split("(")[0]
        key = (short, int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = [f"== rocprofv3 --pmc SQ_* / GRBM_* --kernel-trace -- nbody-bench --gpu {sizes} --steps 100 --warmup 10 --dt 0.01 ==",
         "(per kernel launch, mean over the launches of that grid; SQ_*_CYCLES and SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles summed over waves)"]
for key in sorted(agg, key=lambda k: (k[1], k[0])):
    c = {k: sum(v) / len(v) for k, v in agg[key].items()}
    n = len(agg[key]["_dur_ns"])
    lines.append("")
    lines.append(f"{key[0]}  grid={key[1]} threads, workgroup={key[2]}  ({n} launches)  mean duration {c['_dur_ns'] / 1e3:.2f} us")
    for k in sorted(c):
        if k != "_dur_ns":
            lines.append(f"    {k:24s} {c[k]:.6g}")
    if "SQ_WAVES" in c and "SQ_WAVE_CYCLES" in c:
        waves = c["SQ_WAVES"]
        lines.append(f"    -> {waves:.0f} waves = {waves / 1024:.2f} per SIMD over the launch; mean wave lifetime "
                     f"{4 * c['SQ_WAVE_CYCLES'] / waves:.0f} cycles")
        if "SQ_WAIT_INST_ANY" in c and "SQ_ACTIVE_INST_VALU" in c:
            lines.append(f"    -> share of wave time: VALU executing {c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.2f}, "
                         f"issue-stalled (SQ_WAIT_INST_ANY) {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f}, "
                         f"parked on s_waitcnt/barrier (SQ_WAIT_ANY) {c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:.2f}")
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024
        lines.append(f"    -> VALU issue utilisation over the kernel's active time (SQ_INSTS_VALU * 26/10 / SIMD-cycles): "
                     f"{c['SQ_INSTS_VALU'] * 26 / 10 / simd_cycles:.2f}")
stats = newest(os.path.join(PROF, "stats", "*", "*_kernel_stats.csv"))
if stats:
    lines.append("")
    lines.append("== rocprofv3 --kernel-trace --stats (same command) ==")
    for r in csv.DictReader(open(stats[0])):
        lines.append(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg_ns={float(r['AverageNs']):12.0f} pct={float(r['Percentage']):6.2f}")
text = "\n".join(lines) + "\n"
open(os.path.join(ROOT, "profiles", f"{tag}_{suffix}.txt"), "w").write(text)
print(text)
