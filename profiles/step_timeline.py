#!/usr/bin/env python3
"""Ordered kernel timeline of one replayed step from a rocprofv3 --kernel-trace database: per position in the step the
kernel's name, median duration and median gap to the previous kernel's end.
usage: step_timeline.py <rocprofv3 output dir> [first-kernel-name-prefix]"""
import glob, os, sqlite3, statistics, sys
db = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True))[0]
# a step starts at the kernel that FOLLOWS the optimiser launch (round 4: step_begin_k rides inside another launch and is no
# longer a kernel of its own); an explicit first-kernel prefix may still be given
first = sys.argv[2] if len(sys.argv) > 2 else None
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
if first:
    starts = [i for i, r in enumerate(rows) if r[0].startswith(first)]
else:
    starts = [i + 1 for i, r in enumerate(rows[:-1]) if r[0].startswith("adam_step_k")]
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
steps = steps[len(steps) // 2:]
from collections import Counter
ln = Counter(b - a for a, b in steps).most_common(1)[0][0]
steps = [(a, b) for a, b in steps if b - a == ln]
print(f"# {len(steps)} replayed steps of {ln} kernels")
tot_d = tot_g = 0.0
for p in range(ln):
    d = statistics.median(rows[a + p][2] - rows[a + p][1] for a, b in steps) / 1e3
    g = statistics.median(rows[a + p][1] - rows[a + p - 1][2] for a, b in steps) / 1e3 if p else 0.0
    tot_d += d; tot_g += g
    print(f"{p:3d} {rows[steps[0][0] + p][0][:60]:60s} dur {d:7.2f}  gap {g:6.2f}  t_end {tot_d + tot_g:8.2f}")
print(f"# sum of durations {tot_d:.1f} us, sum of gaps {tot_g:.1f} us")
