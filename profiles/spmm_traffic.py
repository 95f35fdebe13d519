#!/usr/bin/env python3
"""Runs ONLY the gather-SpMM of the timed path — gcn_aggregate_gather_head5_k<32>: Â·[X | ind] straight from the
resident products-scale feature matrix (N = 2,449,029 rows of 400 B) — on a hop-2-shaped frontier
(n = 37.5k destination rows, e = 38k edges from 512 source rows), for the rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE need separate passes on gfx950).
    python profiles/spmm_traffic.py                  # the kernel, 10 launches
    python profiles/spmm_traffic.py --parse D1 D2    # D1/D2: rocprofv3 output dirs of the FETCH / WRITE passes
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N, F, IND = 2_449_029, 100, 4
n_rows, m_src, e_target = 37500, 512, 38000


def algorithmic_bytes(n, e, f):
    return 4 * ((e + n) * f + n * f + (e + n) + (n + 1) + n + f)


def run():
    import numpy as np
    import torch
    from grapes_amd import ops
    rng = np.random.default_rng(0)
    w = rng.pareto(1.2, m_src) + 1
    deg = np.maximum(1, (w / w.sum() * e_target).astype(np.int64))
    srcs = np.sort(rng.permutation(n_rows)[:m_src])
    src = np.repeat(srcs, deg)
    dst = np.concatenate([np.sort(rng.permutation(n_rows)[:d]) for d in deg])
    ls, ld = torch.from_numpy(src).to("cuda", torch.int32), torch.from_numpy(dst).to("cuda", torch.int32)
    ids = torch.from_numpy(np.sort(rng.permutation(N)[:n_rows])).to("cuda", torch.int32)   # ascending global ids
    prep = ops.PreparedGraph(ls, ld, n_rows, src_grouped=True, items_fwd=False, head_ids=ids)   # as the step builds it
    e = int(prep.rowptr_t[n_rows].item())
    X = torch.randn(N, F, device="cuda")
    code = torch.zeros(N, dtype=torch.int32, device="cuda")
    out = torch.empty(n_rows, F + IND, device="cuda")
    for _ in range(10):
        ops.gcn_aggregate_gather(X, ids, prep, code, 1, IND, out=out)
    torch.cuda.synchronize()
    print(json.dumps(dict(n=n_rows, e=e, f=F + IND, algorithmic_bytes=algorithmic_bytes(n_rows, e, F + IND))))


def parse(d_fetch, d_write):
    def counter(d, name):
        vals = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gcn_aggregate_gather" in r["Kernel_Name"] and r["Counter_Name"] == name:
                    vals.append(float(r["Counter_Value"]))
        return vals
    fe, wr = counter(d_fetch, "FETCH_SIZE"), counter(d_write, "WRITE_SIZE")
    fetch_kib, write_kib = sum(fe) / len(fe), sum(wr) / len(wr)
    # MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide
    # (16 B/lane) coalesced reads -> doubled; WRITE_SIZE is exact for 16 B/lane stores.  The gathered rows are
    # 400 B at arbitrary 16 B-aligned offsets (4-5 128-B lines each), so line granularity adds up to ~1.3x to
    # the 400 B/row a byte count would predict; the guide calls other access shapes "uncalibrated".
    read_b, write_b = 2.0 * fetch_kib * 1024.0, write_kib * 1024.0
    alg = algorithmic_bytes(n_rows, 37750, F + IND)
    res = dict(kernel="gcn_aggregate_gather_head5_k<32>", launches=len(fe), fetch_size_kib_raw=fetch_kib,
               write_size_kib_raw=write_kib, hbm_read_bytes_per_launch=read_b, hbm_write_bytes_per_launch=write_b,
               hbm_bytes_per_launch=read_b + write_b,
               correction="read bytes = 2 x FETCH_SIZE x 1024 (gfx950 wide-read half-count), write bytes = WRITE_SIZE x 1024",
               shape=dict(n=n_rows, e=38000, f=F + IND, N=N), algorithmic_bytes=alg)
    out = os.path.join(ROOT, "gpurun_out", "traffic_gcn_aggregate.json")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--parse":
        parse(sys.argv[2], sys.argv[3])
    else:
        run()
