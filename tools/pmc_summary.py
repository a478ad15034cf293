#!/usr/bin/env python3
"""Summarise rocprofv3 output directories for the round profile.

    pmc_summary.py stats  <dir>   -> per-kernel count / total / average duration from *_kernel_trace.csv
    pmc_summary.py pmc    <dir>   -> per-kernel, per-counter sum and per-launch mean from *_counter_collection.csv

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1024 B; on gfx950 FETCH_SIZE counts
128-B requests as 64 B (MI355X_MICROARCH.md, HBM section), so the corrected column doubles it.
"""
import csv, glob, json, os, sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def stats(d):
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {k: {"calls": len(v), "total_ms": sum(v) / 1e6, "avg_ms": sum(v) / len(v) / 1e6, "max_ms": max(v) / 1e6} for k, v in rows.items()}
    return dict(sorted(out.items(), key=lambda kv: -kv[1]["total_ms"]))


def pmc(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        out[k] = {}
        for c, v in cs.items():
            e = {"launches": len(v), "sum": sum(v), "per_launch": sum(v) / len(v)}
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                scale = 2.0 if c == "FETCH_SIZE" else 1.0
                e["bytes_per_launch_corrected"] = e["per_launch"] * 1024.0 * scale
            out[k][c] = e
    return out


if __name__ == "__main__":
    mode, d = sys.argv[1], sys.argv[2]
    print(json.dumps(stats(d) if mode == "stats" else pmc(d), indent=1))
