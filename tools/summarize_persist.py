#!/usr/bin/env python3
"""Condense gpurun_out/prof_persist/ (tools/profile_persist.sh) into one table: per size and setting the step kernel's
mean duration, waves per launch, mean wave lifetime in cycles and VALU issue utilisation over the launch."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PROF = os.path.join(ROOT, "gpurun_out", "prof_persist")
for d in sorted(glob.glob(os.path.join(PROF, "n*_p*"))):
    if not os.path.isdir(d):
        continue
    cc = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    tr = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))
    if not cc or not tr:
        print(os.path.basename(d), "no data")
        continue
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        if "step_kernel" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    top = max(dur, key=lambda k: len(dur[k]))
    cnt = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if r["Kernel_Name"] == top:
            cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
    mean = {k: sum(v) / len(v) for k, v in cnt.items()}
    us = sum(dur[top][10:]) / max(len(dur[top][10:]), 1) / 1e3
    waves = mean.get("SQ_WAVES", 0)
    life = 4 * mean.get("SQ_WAVE_CYCLES", 0) / waves if waves else 0     # SQ_WAVE_CYCLES counts quad-cycles
    busy = mean.get("SQ_BUSY_CYCLES", 0)
    # issue utilisation: VALU instructions x 2.6 cycles of this mix / (1024 SIMDs x launch cycles); launch cycles from
    # SQ_BUSY_CYCLES summed over the 32 shader engines' counters
    print(f"{os.path.basename(d):12s} {top[:60]:60s} launches={len(dur[top]):4d} mean {us:7.2f} us  waves/launch {waves:7.0f}  "
          f"wave lifetime {life:8.0f} cycles  SQ_INSTS_VALU {mean.get('SQ_INSTS_VALU', 0):.3e}  SQ_BUSY_CYCLES {busy:.3e}")
