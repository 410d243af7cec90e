#!/usr/bin/env python3
"""Condense gpurun_out/prof_c23/ (tools/profile_c2_c3.sh) into profiles/<tag>_c2_c3_summary.txt: for BASELINE.json's
configs 2 and 3, which kernels a step launches with every knob on auto, their mean duration by rocprofv3, the roofline
fraction recomputed from that mean (14 flop per interaction against 157.3 TFLOP/s), VALU issue utilisation from the PMC
passes, and the harness' own wall-clock row beside it."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PROF = os.path.join(ROOT, "gpurun_out", "prof_c23")
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
PEAK, FLOP = 157.3e12, 14.0
# mass_len of srand(11037) MakeGalaxies(N, 2) (SURVEY.md section 8 probe; the harness prints N x mass_len / time)
WORLD = {"c2": (65536, 32641), "c3": (262144, 130916), "c3h": (262144, 130916)}
TITLE = {"c2": "C2  N = 65 536, dt = 0.01: 10 warm-up steps + ONE 100-step call (reference harness shape, bench.c:21-35)",
         "c3": "C3  N = 262 144, dt = 0.01: 20-step chain, warm-up call (plain launches) + 3 timed calls (cached hipGraph from the 2nd use)",
         "c3h": "C3  the same chain at dt = 0.005 (dt halved)"}


def newest(pattern):
    best = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


def short(name):
    tail = name.split("(anonymous namespace)::", 1)[-1]
    return tail.split("(")[0]


lines = ["Kernels BASELINE.json's configs 2 and 3 launch with every knob on auto, profiled through the C harness",
         "(tools/profile_c2_c3.sh; rocprofv3 with the program itself after `--`).  frac = interactions per launch x 14 flop /",
         "mean launch duration / 157.3 TFLOP/s -- the same convention as bench.py's roofline.frac, recomputable from this file.",
         "step_kernel<K, W, VARIANT, FUSED, PERSIST>: K receivers per lane, W waves per workgroup, VARIANT 1 = scalar-cache route."]
for cfg in ("c2", "c3", "c3h"):
    d = os.path.join(PROF, cfg)
    if not os.path.isdir(d):
        continue
    n, m = WORLD[cfg]
    lines += ["", "=" * 118, TITLE[cfg], "command: " + open(os.path.join(d, "command.txt")).read().strip()]
    run = open(os.path.join(d, "plain_run.txt")).read().strip().splitlines()
    row = [l.split() for l in run if l.split() and l.split()[0] == str(n)]
    wall_frac = None
    if row:
        r = row[0]
        wall_frac = float(r[3]) / 100.0
        lines.append(f"unprofiled run of the same command: {r[1]} us/step, {r[2]} interactions/s, {r[3]} % of peak by the wall clock of the timed call")
    tr = newest(os.path.join(d, "stats", "*", "*_kernel_trace.csv"))
    if tr:
        rows = list(csv.DictReader(open(tr[0])))
        per = collections.defaultdict(list)
        for r in rows:
            key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"] or 1), int(r["Workgroup_Size_X"]),
                   int(r["LDS_Block_Size"]), int(r["VGPR_Count"]), int(r["SGPR_Count"]), int(r["Scratch_Size"]))
            per[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        lines.append("rocprofv3 --kernel-trace --stats, per (kernel, grid):")
        for key, durs in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            name, grid, wg, lds, vgpr, sgpr, scratch = key
            mean = sum(durs) / len(durs)
            lines.append(f"  {name:44s} x{len(durs):4d}  grid={grid} wg={wg} LDS={lds} VGPR={vgpr} SGPR={sgpr} scratch={scratch}  "
                         f"mean {mean / 1e3:9.2f} us  min {min(durs) / 1e3:9.2f}  max {max(durs) / 1e3:9.2f}")
        steps = [(k, v) for k, v in per.items() if k[0].startswith("step_kernel")]
        launches = sum(len(v) for _, v in steps)
        fin = sum(len(v) for k, v in per.items() if k[0].startswith("finish_kernel"))
        step_ns = sum(sum(v) for _, v in steps)
        if launches:
            mean = step_ns / launches
            inter = float(n) * float(m)
            total_steps = launches   # these worlds fit one source pass: one step-kernel launch per step
            lines.append(f"  -> one step = 1 step_kernel launch" + (" + 1 finish_kernel (two-kernel form: the sources are split and the world has > 200 000 receivers)"
                                                                     if fin else " and nothing else (no source split at this size: no finish kernel, no fused finish)"))
            lines.append(f"  -> interactions per launch = N x mass_len = {n} x {m} = {inter:.4e}; x 14 flop = {inter * FLOP:.4e} flop")
            frac = inter * FLOP / (mean * 1e-9) / PEAK
            best = min(min(v) for _, v in steps)
            lines.append(f"  -> step_kernel mean {mean / 1e3:.2f} us over {total_steps} launches  =>  {inter * FLOP / (mean * 1e-9) / 1e12:.1f} TFLOP/s  =>  frac {frac:.3f}"
                         f"   (fastest launch {best / 1e3:.2f} us => {inter * FLOP / (best * 1e-9) / PEAK:.3f})"
                         + (f"; wall-clock frac of the timed call {wall_frac:.3f}" if wall_frac is not None else ""))
            if cfg == "c2":
                first = [d_ for _, v in steps for d_ in v]
                lines.append(f"     the first 10 launches are the warm-up call (clock ramp after host-side set-up: profiles/r03_c2_first_call_probe.txt): "
                             f"mean of launches 1-10 {sum(first[:10]) / 10e3:.2f} us, of launches 11-110 {sum(first[10:]) / max(len(first) - 10, 1) / 1e3:.2f} us")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "step_kernel" not in r["Kernel_Name"]:
                continue
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in agg.items():
        c = {k: sum(v) / len(v) for k, v in cs.items()}
        lines.append(f"PMC passes (own runs, per launch, mean) of {name}:")
        lines.append("  " + "  ".join(f"{k}={c[k]:.6g}" for k in sorted(c)))
        if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
            simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024
            lines.append(f"  -> VALU issue utilisation = SQ_INSTS_VALU x (26 cycles / 10 instructions of this mix) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = "
                         f"{c['SQ_INSTS_VALU'] * 2.6 / simd_cycles:.3f}")
            waves_inter = float(n) * float(m) / 64.0
            lines.append(f"  -> SQ_INSTS_VALU per wave-interaction = {c['SQ_INSTS_VALU'] / waves_inter:.2f} (10 in the loop body; the rest is prologue, reduction and integrator)")
        if "SQ_INSTS_VALU_TRANS" in c:
            lines.append(f"  -> transcendental share of VALU instructions = {c['SQ_INSTS_VALU_TRANS'] / c['SQ_INSTS_VALU']:.3f} (1 v_rsq_f32 in 10)")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            alg = n * (12 + 8) + m * 12 + n * 24
            lines.append(f"  -> memory side per launch: FETCH_SIZE {c['FETCH_SIZE']:.0f} KiB raw, WRITE_SIZE {c['WRITE_SIZE']:.0f} KiB; algorithmic {alg / 1024:.0f} KiB "
                         f"(reads N x 20 + M x 12, writes N x 24): HBM does not bound this kernel")
text = "\n".join(lines) + "\n"
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", f"{tag}_c2_c3_summary.txt"), "w").write(text)
print(text)
