#!/usr/bin/env python3
"""Writes profiles/bench_static.json — the two per-launch quantities of bench.py's roofline kernel that need their own
rocprofv3 runs of THE SAME COMMAND (python bench.py ...):

  dispatch_ramp_us  = (rocprofv3 --kernel-trace average duration of the gather-SpMM kernel over all its launches)
                      - (bench.py's in-kernel stamp average over the same launches, roofline.avg_launch_us_all)
  traffic           = HBM bytes per launch of that kernel from two --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one
                      pass on gfx950), corrected as MI355X_MICROARCH.md "HBM" prescribes: counters are KiB,
                      read bytes = 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B), write bytes = WRITE_SIZE; averaged
                      over the frontier-sized launches (positions 1 and 2 of the four launches of a step) and over all four.

usage: make_bench_static.py <bench_line.json> <kernel-trace dir> <pmc FETCH dir> <pmc WRITE dir> <commit> [out.json]"""
import csv
import glob
import json
import os
import sqlite3
import sys

KERNEL = "gcn_aggregate_gather_head5_k"


def trace_avg_us(d):
    """average duration (us) of KERNEL over all its dispatches in a --kernel-trace run (csv stats or rocpd database)"""
    for f in sorted(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Name"]:
                return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    for f in sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)):
        v = [float(x[0]) for x in sqlite3.connect(f).execute("select duration from kernels where name like ?", (f"%{KERNEL}%",))]
        if v:
            return sum(v) / len(v) / 1e3, len(v)
    raise SystemExit(f"{KERNEL}: no kernel-trace rows under {d}")


def counter_per_dispatch(d, name):
    """[value per dispatch of KERNEL, in dispatch order] for counter `name`"""
    rows = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name:
                rows.append((int(r.get("Dispatch_Id", len(rows))), float(r["Counter_Value"])))
    if not rows:
        for f in sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)):
            con = sqlite3.connect(f)
            try:
                q = con.execute("select dispatch_id, value from counters_collection where kernel_name like ? and counter_name = ?",
                                (f"%{KERNEL}%", name))
                rows += [(int(a), float(b)) for a, b in q]
            except sqlite3.Error:
                pass
    rows.sort()
    return [v for _, v in rows]


def main():
    line_f, d_trace, d_fetch, d_write, commit = sys.argv[1:6]
    out = sys.argv[6] if len(sys.argv) > 6 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_static.json")
    line = json.load(open(line_f))
    roof = line["roofline"]
    t_us, calls = trace_avg_us(d_trace)
    stamp_all = roof.get("avg_launch_us_stamps", roof.get("avg_launch_us_all"))
    res = dict(commit=commit, command="python bench.py (products workload, defaults)", kernel=KERNEL,
               kernel_trace_avg_us=round(t_us, 3), kernel_trace_calls=calls, stamp_avg_us_all_positions=stamp_all,
               dispatch_ramp_us=round(max(t_us - stamp_all, 0.0), 3))
    fe, wr = counter_per_dispatch(d_fetch, "FETCH_SIZE"), counter_per_dispatch(d_write, "WRITE_SIZE")
    if fe and wr:
        npos = roof.get("launches_per_step_all", roof.get("launches_per_step", 4))

        def per_pos(v):
            # the LAST replayed steps only: the run begins with eager warm-up steps and the capture, whose launches of the kernel
            # (all of them on their own there) are not the timed step's
            keep = npos * 6 if len(v) >= npos * 8 else len(v) - len(v) % npos
            v = v[len(v) - keep:]
            return [sum(v[p::npos]) / max(1, len(v[p::npos])) for p in range(npos)]
        fp, wp = per_pos(fe), per_pos(wr)
        big = sorted(range(npos), key=lambda p: -(fp[p] + wp[p]))[:roof.get("launches_per_step_frontier", 2)]
        rd = lambda kib: 2.0 * kib * 1024.0
        wb = lambda kib: kib * 1024.0
        read_b = sum(rd(fp[p]) for p in big) / len(big)
        write_b = sum(wb(wp[p]) for p in big) / len(big)
        # the algorithmic bytes of the launches the counters were collected ON: the counter passes are short runs of the command
        # (a dozen steps), whose frontiers are those of an untrained sampler — bench.py prints the last step's in its line
        alg, alg_src = roof["avg_algorithmic_bytes"], "the bench line's roofline probe (another training state than the counter passes)"
        for lf in sorted(glob.glob(os.path.join(d_fetch, "..", "pmc_f.log"))) + sorted(glob.glob(os.path.join(os.path.dirname(d_fetch.rstrip("/")), "pmc_f.log"))):
            try:
                pl = [json.loads(x) for x in open(lf) if x.startswith("{")][-1]
                lg = sorted((g["bytes"] for g in pl["config"]["last_step_gather"]), reverse=True)[:len(big)]
                alg, alg_src = int(sum(lg) / len(lg)), "last timed step of the FETCH_SIZE counter pass (config.last_step_gather of its line)"
                break
            except Exception:
                continue
        res["traffic"] = dict(
            hbm_bytes_per_launch=int(read_b + write_b), hbm_read_bytes_per_launch=int(read_b), hbm_write_bytes_per_launch=int(write_b),
            basis="frontier-sized launches (the %d largest of the %d launches of a step)" % (len(big), npos),
            per_position=[dict(position=p, read_bytes=int(rd(fp[p])), write_bytes=int(wb(wp[p]))) for p in range(npos)],
            dispatches=dict(fetch=len(fe), write=len(wr)),
            correction="counters in KiB; read bytes = 2 x FETCH_SIZE x 1024 (gfx950 half-count of wide reads), write bytes = "
                       "WRITE_SIZE x 1024 (MI355X_MICROARCH.md, HBM); separate --pmc passes of `python bench.py`",
            algorithmic_bytes_per_launch=alg, algorithmic_bytes_source=alg_src, unique_bytes_per_launch=roof.get("avg_unique_bytes"),
            ratio_to_algorithmic=round((read_b + write_b) / alg, 3),
            ratio_to_unique=(round((read_b + write_b) / roof["avg_unique_bytes"], 3) if roof.get("avg_unique_bytes") else None),
            commit=commit)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
