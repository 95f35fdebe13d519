#!/usr/bin/env python3
"""Idle time inside and between graph replays, from a rocprofv3 --kernel-trace database.
usage: replay_gaps.py <rocprofv3 output dir>"""
import glob, os, sqlite3, statistics, sys
db = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True))[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
print("columns:", cols)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
# a step starts at indicator_mark_k (first kernel of the captured step)
starts = [i for i, r in enumerate(rows) if r[0].startswith("indicator_mark_k")]
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
steps = steps[len(steps) // 2:]                      # the timed, replayed half
intra, inter, busy, span = [], [], [], []
for a, b in steps:
    ks = rows[a:b]
    # trailing eager kernels of the bench loop (the edge-count accumulation) belong to the step's period
    busy.append(sum(k[2] - k[1] for k in ks))
    span.append(rows[b][1] - ks[0][1])
    gaps = [ks[i + 1][1] - ks[i][2] for i in range(len(ks) - 1)]
    intra.append(sum(g for g in gaps if g > 0))
    inter.append(rows[b][1] - ks[-1][2])
print(f"steps {len(steps)}: period {statistics.mean(span)/1e3:.1f} us, kernels busy {statistics.mean(busy)/1e3:.1f} us, "
      f"gaps inside a step {statistics.mean(intra)/1e3:.1f} us, gap to the next step's first kernel {statistics.mean(inter)/1e3:.1f} us")
