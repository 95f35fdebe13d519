#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run: per-kernel calls / total / avg / median, launches per step.
Reads either the csv output (*_kernel_stats.csv) or the rocpd sqlite database (*_results.db).
usage: summarize_rocprof.py <output dir of rocprofv3> [steps]"""
import csv, glob, os, sqlite3, statistics, sys

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
rows = []                               # (name, calls, total_ns, avg_ns, median_ns, min_ns, max_ns)
csvs = sorted(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True))
dbs = sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True))
if csvs:
    src = csvs[0]
    for r in csv.DictReader(open(src)):
        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float("nan"),
                     float(r["MinNs"]), float(r["MaxNs"])))
elif dbs:
    src = dbs[0]
    per = {}
    for name, dur in sqlite3.connect(src).execute("select name, duration from kernels"):
        per.setdefault(name, []).append(float(dur))
    for name, v in per.items():
        rows.append((name, len(v), sum(v), sum(v) / len(v), statistics.median(v), min(v), max(v)))
    rows.sort(key=lambda r: -r[2])
else:
    sys.exit(f"no *_kernel_stats.csv or *_results.db under {d}")
tot = sum(r[2] for r in rows)
calls = sum(r[1] for r in rows)
print(f"# {os.path.basename(src)}\n# total kernel time {tot/1e6:.3f} ms over {calls} launches"
      + (f"; per step: {tot/1e6/steps:.3f} ms, {calls/steps:.0f} launches" if steps else ""))
print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'med_us':>8s} {'min_us':>8s} {'max_us':>9s} {'pct':>6s}")
for name, c, t, a, med, lo, hi in rows[:48]:
    short = name.replace("void ", "").split("(")[0][:72]
    print(f"{short:72s} {c:7d} {t/1e6:10.3f} {a/1e3:9.2f} {med/1e3:8.2f} {lo/1e3:8.2f} {hi/1e3:9.2f} {100*t/tot:6.2f}")
