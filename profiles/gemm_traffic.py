#!/usr/bin/env python3
"""Runs ONLY the two bf16x3 GEMMs of the aggregate-first layers at the hop-2 shape (n = 37.5k rows, 104 -> 256) for the
rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate passes on gfx950).
    python profiles/gemm_traffic.py                  # the kernels, 10 launches each
    python profiles/gemm_traffic.py --parse D1 D2    # D1/D2: rocprofv3 output dirs of the FETCH / WRITE passes
"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n, fi, fo = 37500, 104, 256


def run():
    import torch
    from grapes_amd import ops
    torch.manual_seed(0)
    x = torch.randn(n, fi, device="cuda"); w = torch.randn(fo, fi, device="cuda") * 0.1; b = torch.randn(fo, device="cuda")
    w2 = torch.randn(1, fo, device="cuda"); rs = torch.randn(n, device="cuda") * 0.1
    dw = torch.empty(fo, fi, device="cuda"); db = torch.empty(fo, device="cuda"); dh = torch.empty(fo, device="cuda")
    for _ in range(10):
        out, head = ops.linear_bias_act_head_fwd(x, w, b, True, w2)
        ops.linear_bwd_weight_gated(None, x, gate=out, dw=dw, dbias=db, accumulate=False, row_scale=rs, col_vec=w2.view(-1), dw_head=dh)
    torch.cuda.synchronize()
    print("ok")


def parse(d_fetch, d_write):
    def counter(d, kern, name):
        vals = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"] and r["Counter_Name"] == name:
                    vals.append(float(r["Counter_Value"]))
        return vals
    res = {}
    for kern, alg_r, alg_w in (("gemm_wsplit_f32_k", 4 * (n * fi + fo * fi + fo), 4 * (n * fo + n)),
                               ("gemm_dw_split_k", 4 * (n * fo + n * fi + n), 4 * 256 * (fo * fi + 2 * fo))):
        fe, wr = counter(d_fetch, kern, "FETCH_SIZE"), counter(d_write, kern, "WRITE_SIZE")
        if not fe or not wr:
            continue
        # MI355X_MICROARCH.md §HBM: KiB units; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads
        read_b, write_b = 2.0 * sum(fe) / len(fe) * 1024.0, sum(wr) / len(wr) * 1024.0
        res[kern] = dict(launches=len(fe), hbm_read_bytes_per_launch=read_b, hbm_write_bytes_per_launch=write_b,
                         algorithmic_read_bytes=alg_r, algorithmic_write_bytes=alg_w,
                         note="read = 2 x FETCH_SIZE x 1024 (gfx950 wide-read half-count), write = WRITE_SIZE x 1024")
    res["shape"] = dict(n=n, f_in=fi, f_out=fo)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "traffic_split_gemms.json"), "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--parse":
        parse(sys.argv[2], sys.argv[3])
    else:
        run()
