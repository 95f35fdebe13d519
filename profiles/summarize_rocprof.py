#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run: per-kernel calls / total / avg, launches per step.
usage: summarize_rocprof.py <dir with *_kernel_stats.csv> [steps]"""
import csv, glob, os, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"# {f}\n# total kernel time {tot/1e6:.3f} ms over {calls} launches" + (f"; per step: {tot/1e6/steps:.3f} ms, {calls/steps:.0f} launches" if steps else ""))
print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'pct':>6s}")
for r in rows[:40]:
    print(f"{r['Name'][:72]:72s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:9.2f} {float(r['MinNs'])/1e3:8.2f} {float(r['MaxNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")
