#!/usr/bin/env python3
"""Summary of profiles/pmc_gather_r05.sh: per counter the mean over the launches of the step's frontier-sized gather-SpMM
(gcn_aggregate_gather_head5_k, and for comparison gemm_wsplit_f32_k and compact_emit_k), with the derived ratios the question needs."""
import collections, csv, glob, sys
root = sys.argv[1]
kern = {"gather": "gcn_aggregate_gather_head5_k", "xw gemm": "gemm_wsplit_f32_k", "compaction": "compact_emit_k", "draw": "sampler_draw_k"}
acc = {k: collections.defaultdict(list) for k in kern}
dur = {k: [] for k in kern}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k, pat in kern.items():
            if pat in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k, pat in kern.items():
            if pat in r["Kernel_Name"]:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in kern:
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    if not c:
        continue
    d = sorted(dur[k]); med = d[len(d) // 2] if d else float("nan")
    print(f"== {kern[k]}   ({len(next(iter(acc[k].values())))} launches per counter; median duration under the profiler {med:.2f} us)")
    for n in sorted(c):
        print(f"   {n:34s} {c[n]:16.1f}")
    g = lambda n: c.get(n, float("nan"))
    wc = g("SQ_WAVE_CYCLES")
    if wc == wc:
        print(f"   -> of the wavefronts' cycles: parked on a wait (s_waitcnt / barrier) {g('SQ_WAIT_ANY') / wc:6.1%}, issue stalls {g('SQ_WAIT_INST_ANY') / wc:6.1%}, "
              f"executing {g('SQ_ACTIVE_INST_ANY') / wc:6.1%} (vector memory {g('SQ_ACTIVE_INST_VMEM') / wc:6.1%})")
        print(f"   -> wavefronts {g('SQ_WAVES'):.0f}, vector-memory read instructions per wavefront {g('SQ_INSTS_VMEM_RD') / g('SQ_WAVES'):.1f}")
    if g("TCC_REQ_sum") == g("TCC_REQ_sum"):
        print(f"   -> L2: {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):6.1%} of {g('TCC_REQ_sum'):.0f} requests hit")
    if g("TCC_EA0_RDREQ_sum") == g("TCC_EA0_RDREQ_sum"):
        print(f"   -> fabric reads {g('TCC_EA0_RDREQ_sum'):.0f} ({g('TCC_EA0_RDREQ_32B_sum'):.0f} of them 32-byte), {g('TCC_EA0_RDREQ_DRAM_sum'):.0f} to DRAM")
    if g("TCP_TCC_READ_REQ_sum") == g("TCP_TCC_READ_REQ_sum") and g("TCP_TCC_READ_REQ_sum") > 0:
        print(f"   -> L1 -> L2 read latency {g('TCP_TCC_READ_REQ_LATENCY_sum') / g('TCP_TCC_READ_REQ_sum'):.0f} cycles per request over {g('TCP_TCC_READ_REQ_sum'):.0f} requests")
