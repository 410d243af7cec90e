#!/usr/bin/env python3
"""Condense gpurun_out/prof/ (tools/profile.sh) into profiles/<tag>_*.{txt,json} + profiles/pmc_traffic.json."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PROF = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)



def newest(pattern):
    """gpurun merges every call's files into gpurun_out/, so a directory can hold several runs: per directory, the file
    written last is the one this summary is about."""
    best = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


lines = []
stats = newest(os.path.join(PROF, "stats", "*", "*_kernel_stats.csv"))
step_avg_ns = None
if stats:
    lines.append("== rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extra-configs --steps 10 --warmup 2 ==")
    for r in csv.DictReader(open(stats[0])):
        lines.append(f"{r['Name'][:90]:90s} calls={r['Calls']:>4s} avg_ns={float(r['AverageNs']):14.0f} total_ns={r['TotalDurationNs']:>14s} pct={float(r['Percentage']):7.3f}")
        if "step_kernel" in r["Name"] and step_avg_ns is None:   # rows are sorted by total time: the headline variant first
            step_avg_ns = float(r["AverageNs"])
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write(open(stats[0]).read())
    tr = newest(os.path.join(PROF, "stats", "*", "*_kernel_trace.csv"))
    if tr:
        rows = [r for r in csv.DictReader(open(tr[0])) if "step_kernel" in r["Kernel_Name"]]
        if rows:
            # bench.py's extra_configs (C2 / C3) launch the same template instantiations at smaller sizes: the figure that must
            # agree with roofline.kernel_ms_per_launch is the mean over the HEADLINE dispatches only -- the most frequent kernel
            # name at its largest grid (N = 2^20)
            top = collections.Counter(r["Kernel_Name"] for r in rows).most_common(1)[0][0]
            big = max(int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) for r in rows if r["Kernel_Name"] == top)
            head = [r for r in rows if r["Kernel_Name"] == top and int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) == big]
            durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in head]
            step_avg_ns = sum(durs) / len(durs)
            r = head[0]
            lines.append(f"headline step kernel {top}: {len(head)} dispatches of grid={r['Grid_Size_X']}x{r.get('Grid_Size_Y', 1)} "
                         f"wg={r['Workgroup_Size_X']} LDS={r['LDS_Block_Size']} VGPR={r['VGPR_Count']} SGPR={r['SGPR_Count']} "
                         f"scratch={r['Scratch_Size']}: mean {step_avg_ns:.0f} ns, min {min(durs)} ns, max {max(durs)} ns "
                         f"(the per-name rows above also hold the smaller launches of extra_configs C2 / C3)")

# the profiled command's own JSON line: its HIP-event figure for the same launches must agree with the trace's mean
_log = os.path.join(PROF, "stats.log")
if os.path.exists(_log):
    for _line in open(_log, errors="replace"):
        if _line.startswith("{") and '"roofline"' in _line:
            _rec = json.loads(_line)
            lines.append(f"the profiled run's own bench line: roofline.kernel_ms_per_launch = {_rec['roofline']['kernel_ms_per_launch']:.3f} ms "
                         f"(HIP events on the launch stream, incl. one ~10 us finish kernel per launch), ms_per_step = {_rec['ms_per_step']:.3f}, "
                         f"frac = {_rec['roofline']['frac']:.4f}")

# bench.py also times the LDS-tile route (roofline.alt_lds): the counters below are those of the HEADLINE variant only,
# i.e. of the step_kernel instantiation with the most launches
files = newest(os.path.join(PROF, "pmc_*", "*", "*_counter_collection.csv"))
names = collections.Counter(r["Kernel_Name"] for f in files for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"])
headline = names.most_common(1)[0][0] if names else None
agg = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"] == headline:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
mean = {k: sum(v) / len(v) for k, v in agg.items()}


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


lines.append("")
lines.append(f"== rocprofv3 --pmc <counters> --kernel-trace, one pass per counter group; per launch (mean) of {headline} ==")
for k in sorted(mean):
    lines.append(f"{k:28s} {mean[k]:.6g}   (n={len(agg[k])})")

summary = {"tag": tag, "step_kernel_avg_ns_stats_pass": step_avg_ns}
if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    dur = mean["_dur_ns"] * 1e-9
    # the MEDIAN launch: one launch in ten can read three times the usual (round 5: 111 MB beside nine launches of 34-35 MB),
    # and a per-launch figure should describe the launch, not that one
    fetch_raw = median(agg["FETCH_SIZE"]) * 1024.0      # rocprofv3 reports KiB
    write = median(agg["WRITE_SIZE"]) * 1024.0
    lines.append("")
    lines.append("FETCH_SIZE per launch, KiB: " + " ".join(f"{v:.0f}" for v in agg["FETCH_SIZE"]) + f"   (median {median(agg['FETCH_SIZE']):.0f}, mean {mean['FETCH_SIZE']:.0f})")
    # MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams;
    # doubling is the prescribed correction and an upper bound for this kernel's 64-B scalar-cache line fills
    fetch_corr = 2.0 * fetch_raw
    summary.update({
        "fetch_bytes_raw": fetch_raw, "fetch_bytes_corrected_x2": fetch_corr, "write_bytes": write,
        "hbm_bytes_per_launch": fetch_corr + write,
        "hbm_GBps": (fetch_corr + write) / dur / 1e9,
        "launch_seconds_pmc_pass": dur,
    })
    lines.append(f"memory-side traffic per launch (median): fetch {fetch_raw/1e6:.1f} MB raw ({fetch_corr/1e6:.1f} MB with the gfx950 x2 correction), "
                 f"write {write/1e6:.1f} MB  ->  {(fetch_corr + write)/dur/1e9:.1f} GB/s of ~8000 GB/s HBM peak")
if "GRBM_GUI_ACTIVE" in mean:
    clk = mean["GRBM_GUI_ACTIVE"] / 8.0 / (mean["_dur_ns"] * 1e-9)
    summary["effective_clock_GHz"] = clk / 1e9
    lines.append(f"effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration): {clk/1e9:.3f} GHz")
if "SQ_INSTS_VALU" in mean:
    n, m = 1 << 20, None
    summary["valu_wave_instructions_per_launch"] = mean["SQ_INSTS_VALU"]
    lines.append(f"SQ_INSTS_VALU per launch {mean['SQ_INSTS_VALU']:.4g} wave-instructions")
if "SQ_ACTIVE_INST_VALU" in mean and "GRBM_GUI_ACTIVE" in mean:
    simd_cycles = mean["GRBM_GUI_ACTIVE"] / 8.0 * 1024      # 1024 SIMDs, cycles each
    busy = 4.0 * mean["SQ_ACTIVE_INST_VALU"] / simd_cycles   # SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)
    summary["valu_busy_fraction"] = busy
    lines.append(f"VALU busy = 4 * SQ_ACTIVE_INST_VALU / (SIMDs * cycles) = {busy:.3f}   (> 1: the counter adds up every wave's "
                 f"in-flight time, and several waves' instructions overlap in the pipe)")
    if "SQ_INSTS_VALU" in mean:
        # the step kernel's mix per interaction: 9 plain (2 cycles) + 1 v_rsq_f32 (8 cycles)
        # = 26 issue cycles per 10 instructions (profiles/r01_ubench4_true_cycles.txt; round 1's packed body: 26 per 8)
        issue = mean["SQ_INSTS_VALU"] * (26.0 / 10.0) / simd_cycles
        summary["valu_issue_utilisation"] = issue
        lines.append(f"VALU issue utilisation = SQ_INSTS_VALU * (26/10 cycles per instruction of this mix) / (SIMDs * cycles) = {issue:.3f}")
if "TCC_HIT_sum" in mean:
    hr = mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
    summary["l2_hit_rate"] = hr
    lines.append(f"L2 hit rate {hr:.4f}")

text = "\n".join(lines) + "\n"
open(os.path.join(out_dir, f"{tag}_rocprof_summary.txt"), "w").write(text)
json.dump(summary, open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w"), indent=1)
if "hbm_bytes_per_launch" in summary:
    sys.path.insert(0, ROOT)
    import bench  # kernel_sources_sha(): bench.py reports the figure only while the kernel sources still hash to this
    # ... and while it launches the shape the profiled run launched (that run's own JSON line is in stats.log)
    launch = None
    log = os.path.join(PROF, "stats.log")
    if os.path.exists(log):
        for line in open(log, errors="replace"):
            if line.startswith("{") and '"roofline"' in line:
                rec = json.loads(line)
                launch = dict(rec["config"]["kernel"], passes=max(rec["roofline"]["launches"] // max(rec["steps"], 1), 1))
    json.dump({"hbm_bytes_per_launch": summary["hbm_bytes_per_launch"], "fetch_bytes_raw": summary["fetch_bytes_raw"],
               "write_bytes": summary["write_bytes"], "source": f"profiles/{tag}_pmc_summary.json",
               "n": 1 << 20, "kernel_sources_sha256": bench.kernel_sources_sha(), "launch": launch,
               "note": "hbm_bytes_per_launch = FETCH_SIZE*1024*2 + WRITE_SIZE*1024, median step_kernel launch of the PMC passes, N=2^20.  The x2 is "
                       "MI355X_MICROARCH.md's gfx950 correction, calibrated on 16-B-per-lane coalesced streaming loads; this "
                       "kernel loads 8-B float2 / 4-B float per lane plus 64-B scalar-cache lines (uncalibrated widths), so the "
                       "corrected figure is an upper bound and fetch_bytes_raw + write_bytes a lower bound.  Run "
                       "tools/summarize_profile.py on the same tree the profile was taken from"},
              open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
print(text)
