#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of one gpurun call (gpurun_out/prof_*) into profiles/.

  python tools/summarize_profiles.py r01

Writes profiles/kernel_stats_<tag>.csv (rocprofv3 --kernel-trace --stats, verbatim),
profiles/drillup_1e8_<tag>.txt (per-dispatch durations of the 10^8-cell launches) and
profiles/traffic_<tag>.json (HBM bytes per launch from the separate --pmc FETCH_SIZE and
--pmc WRITE_SIZE passes; gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE reports half of
the bytes of a 16 B/lane coalesced stream, both counters are in KiB)."""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
BIG_GRID = 2500096  # ceil(10^8/10/4/256) workgroups x 256 lanes



def newest(pattern):
    """gpurun merges every call's files into gpurun_out/, so earlier runs' outputs linger: take the latest."""
    return max(glob.glob(pattern), key=os.path.getmtime)


os.makedirs(PROF, exist_ok=True)
stats = newest(os.path.join(OUT, "prof_kt", "*", "*_kernel_stats.csv"))
shutil.copy(stats, os.path.join(PROF, "kernel_stats_%s.csv" % tag))
trace = newest(os.path.join(OUT, "prof_kt", "*", "*_kernel_trace.csv"))
rows = list(csv.DictReader(open(trace)))
# bench.py issues the headline launches first: warmup + steps (timed loop) + steps (kernel-only loop);
# later launches of the same kernel (config 3 / config 5 extras) can share the grid size, so only
# the first `HEAD` dispatches of each kernel name are the 10^8-cell drillUp
HEAD_KT, HEAD_PMC = 10 + 2 * 100, 2 + 2 * 20
big = {}
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the headline kernel instantiation = the first 10^8-cell drillUp launch bench.py issues
HEADNAME = next(r["Kernel_Name"] for r in rows if int(r["Grid_Size_X"]) == BIG_GRID and "drillup_rows_kernel" in r["Kernel_Name"])
for r in rows:
    if int(r["Grid_Size_X"]) == BIG_GRID and "drillup_rows_kernel" in r["Kernel_Name"]:
        lst = big.setdefault(r["Kernel_Name"], [])
        if len(lst) < (HEAD_KT if r["Kernel_Name"] == HEADNAME else 10 ** 9):
            lst.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = ["rocprofv3 --kernel-trace, dispatches of the 10^8-cell drillUp (grid %d lanes), ns" % BIG_GRID]
summary = {}
for k, v in big.items():
    lines.append("%s\n  calls=%d  mean=%.0f  median=%.0f  min=%d  max=%d" % (k, len(v), statistics.mean(v), statistics.median(v), min(v), max(v)))
    summary[k] = statistics.mean(v)
open(os.path.join(PROF, "drillup_1e8_%s.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))


def counter(dirname, name):
    f = newest(os.path.join(OUT, dirname, "*", "*_counter_collection.csv"))
    per = {}
    recs = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in recs:
        if r["Counter_Name"] == name and int(r["Grid_Size"]) == BIG_GRID:
            lst = per.setdefault(r["Kernel_Name"], [])
            if len(lst) < (HEAD_PMC if r["Kernel_Name"] == HEADNAME else 10 ** 9):
                lst.append(float(r["Counter_Value"]))
    return {k: statistics.median(v) for k, v in per.items()}


fetch, write = counter("prof_fetch", "FETCH_SIZE"), counter("prof_write", "WRITE_SIZE")
traffic = {}
for k in fetch:
    traffic[k] = {"FETCH_SIZE_KiB": fetch[k], "WRITE_SIZE_KiB": write.get(k),
                  "hbm_bytes": 2 * fetch[k] * 1024 + write.get(k, 0) * 1024}
head = [k for k in traffic if k == HEADNAME]
doc = {"note": "HBM bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (separate --pmc passes, gfx950 FETCH_SIZE x2 correction)",
       "kernels": traffic, "hbm_bytes_per_launch": traffic[head[0]]["hbm_bytes"] if head else None}
json.dump(doc, open(os.path.join(PROF, "traffic_%s.json" % tag), "w"), indent=1)
print(json.dumps(doc, indent=1))
