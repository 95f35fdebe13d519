#!/usr/bin/env python3
"""Runs ONLY the forward gather-SpMM (gcn_aggregate_k<4>) on a north-star-shaped frontier graph, for the
rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate passes on gfx950).  The frontier is the
hop-2 shape of bench.py: n = 37.5k rows, e = 38k edges from 512 source rows, F = 256.
    python profiles/spmm_traffic.py            # prints algorithmic bytes per launch
    python profiles/spmm_traffic.py --parse D1 D2   # D1/D2: rocprofv3 output dirs of the two pmc passes
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def algorithmic_bytes(n, e, f):
    return 4 * ((e + n) * f + n * f + (e + n) + (n + 1) + n + f)


def run():
    import numpy as np
    import torch
    from grapes_amd import ops
    rng = np.random.default_rng(0)
    n, m, e_target, H = 37500, 512, 38000, 256
    w = rng.pareto(1.2, m) + 1
    deg = np.maximum(1, (w / w.sum() * e_target).astype(np.int64))
    srcs = np.sort(rng.permutation(n)[:m])
    src = np.repeat(srcs, deg)
    dst = np.concatenate([np.sort(rng.permutation(n)[:d]) for d in deg])
    ls, ld = torch.from_numpy(src).to("cuda", torch.int32), torch.from_numpy(dst).to("cuda", torch.int32)
    prep = ops.PreparedGraph(ls, ld, n, src_grouped=True, items_fwd=False)
    e = int(prep.rowptr_t[n].item())
    h = torch.randn(n, H, device="cuda")
    b = torch.randn(H, device="cuda")
    out = torch.empty_like(h)
    for _ in range(10):
        ops.gcn_aggregate_fwd(h, prep, b, True, out=out)
    torch.cuda.synchronize()
    print(json.dumps(dict(n=n, e=e, f=H, algorithmic_bytes=algorithmic_bytes(n, e, H))))


def parse(d_fetch, d_write):
    def counter(d, name):
        vals = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gcn_aggregate_k" in r["Kernel_Name"] and r["Counter_Name"] == name:
                    vals.append(float(r["Counter_Value"]))
        return vals
    fe, wr = counter(d_fetch, "FETCH_SIZE"), counter(d_write, "WRITE_SIZE")
    fetch_kib = sum(fe) / len(fe)
    write_kib = sum(wr) / len(wr)
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of
    # the bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE is exact for 16 B/lane stores.
    read_b = 2.0 * fetch_kib * 1024.0
    write_b = write_kib * 1024.0
    res = dict(kernel="gcn_aggregate_k<4>", launches=len(fe), fetch_size_kib_raw=fetch_kib, write_size_kib_raw=write_kib,
               hbm_read_bytes_per_launch=read_b, hbm_write_bytes_per_launch=write_b,
               hbm_bytes_per_launch=read_b + write_b,
               correction="read bytes = 2 x FETCH_SIZE x 1024 (gfx950 wide-read half-count), write bytes = WRITE_SIZE x 1024",
               shape=dict(n=37500, e=38000, f=256), algorithmic_bytes=algorithmic_bytes(37500, 37750, 256))
    json.dump(res, open(os.path.join(ROOT, "profiles", "traffic_gcn_aggregate.json"), "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--parse":
        parse(sys.argv[2], sys.argv[3])
    else:
        run()
