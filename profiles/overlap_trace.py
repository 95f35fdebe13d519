#!/usr/bin/env python3
"""Do kernels of different queues overlap in a rocprofv3 --kernel-trace database?  Prints, for the second half of the trace,
per queue the kernel count and busy time, the union busy time, and the time during which >= 2 kernels were running.
usage: overlap_trace.py <rocprofv3 output dir>"""
import glob, os, sqlite3, sys
db = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True))[0]
con = sqlite3.connect(db)
rows = con.execute("select queue_id, stream_id, start, end, name from kernels order by start").fetchall()
rows = rows[len(rows) // 2:]
t0, t1 = rows[0][2], max(r[3] for r in rows)
per = {}
for q, s, a, b, n in rows:
    c = per.setdefault((q, s), [0, 0])
    c[0] += 1; c[1] += b - a
ev = sorted([(a, 1) for _, _, a, b, _ in rows] + [(b, -1) for _, _, a, b, _ in rows])
busy = two = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: two += t - last
    depth += d; last = t
print(f"window {(t1 - t0) / 1e3:.1f} us, {len(rows)} kernels")
for k, (cnt, dur) in sorted(per.items()):
    print(f"  queue {k[0]} stream {k[1]}: {cnt} kernels, {dur / 1e3:.1f} us of kernel time")
print(f"  union busy {busy / 1e3:.1f} us; >= 2 kernels running for {two / 1e3:.1f} us")
