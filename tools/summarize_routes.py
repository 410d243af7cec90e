#!/usr/bin/env python3
"""Condense gpurun_out/prof_routes/ (tools/profile_routes.sh) into profiles/<tag>_routes_pmc.txt: the step kernel's
counters per launch for the scalar-cache route (variant 1) next to the LDS-tile route (variant 0)."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PROF = os.path.join(ROOT, "gpurun_out", "prof_routes")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def newest(pattern):
    """per directory, the file written last (gpurun merges every call's files into gpurun_out/)"""
    best = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


def collect(v):
    mean, kernel = {}, None
    files = newest(os.path.join(PROF, f"v{v}_pmc_*", "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if "step_kernel" not in r["Kernel_Name"]:
                continue
            kernel = r["Kernel_Name"]
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, vals in agg.items():
        mean[k] = sum(vals) / len(vals)
    stats = newest(os.path.join(PROF, f"v{v}_stats", "*", "*_kernel_stats.csv"))
    if stats:
        for r in csv.DictReader(open(stats[0])):
            if "step_kernel" in r["Name"]:
                mean["_avg_ns_stats_pass"] = float(r["AverageNs"])
                break
    tr = newest(os.path.join(PROF, f"v{v}_stats", "*", "*_kernel_trace.csv"))
    if tr:
        for r in csv.DictReader(open(tr[0])):
            if "step_kernel" in r["Kernel_Name"]:
                mean["_vgpr"], mean["_sgpr"], mean["_lds"] = float(r["VGPR_Count"]), float(r["SGPR_Count"]), float(r["LDS_Block_Size"])
                break
    return kernel, mean


k1, smem = collect(1)
k0, lds = collect(0)
lines = [f"== tools/profile_routes.sh: rocprofv3 --pmc passes of `bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1`,",
         f"   N = 2^20, per step-kernel launch (mean); quad-cycle counters (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*) as reported ==",
         f"scalar-cache route: {k1}", f"LDS-tile route:     {k0}", "",
         f"{'counter':30s} {'scalar cache (1)':>18s} {'LDS tiles (0)':>18s} {'LDS / scalar':>14s}"]
for key in sorted(set(smem) | set(lds)):
    a, b = smem.get(key), lds.get(key)
    ratio = f"{b / a:14.3f}" if a and b else f"{'':14s}"
    fa = f"{a:18.6g}" if a is not None else f"{'-':>18s}"
    fb = f"{b:18.6g}" if b is not None else f"{'-':>18s}"
    lines.append(f"{key:30s} {fa} {fb} {ratio}")


def derived(m, name):
    out = []
    if "GRBM_GUI_ACTIVE" in m and "_dur_ns" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        out.append(f"{name}: shader cycles per launch {cyc:.4g} (clock {cyc / m['_dur_ns']:.3f} GHz in the PMC pass)")
        simd = cyc * 1024
        if "SQ_INSTS_VALU" in m:
            out.append(f"{name}: VALU issue utilisation (SQ_INSTS_VALU x 26/10 cycles / SIMD-cycles) {m['SQ_INSTS_VALU'] * 2.6 / simd:.3f}")
        if "SQ_WAVE_CYCLES" in m:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                      "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"):
                if c in m:
                    out.append(f"{name}: {c} / SQ_WAVE_CYCLES = {m[c] / m['SQ_WAVE_CYCLES']:.4f}")
    return out


lines.append("")
lines += derived(smem, "scalar")
lines += derived(lds, "LDS   ")
text = "\n".join(lines) + "\n"
open(os.path.join(ROOT, "profiles", f"{tag}_routes_pmc.txt"), "w").write(text)
print(text)
