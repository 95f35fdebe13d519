# world-size-1 cost of the multi-GPU paths on one MI355X (same box, back to back):
#   single | peer-mapped shards (8 in-process shards + RCCL gradient all-reduce) | RCCL halo exchange (--force_partition)
set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/peer
O=gpurun_out/peer
A="--steps 500 --warmup 100 --cpu_steps 0 --eager_steps 0 --no_roofline"
timeout -k 10 200 python bench.py $A > $O/single.json 2> $O/single.err || { tail -5 $O/single.err; exit 1; }
timeout -k 10 200 python bench.py $A --force_peer --peer_shards 8 > $O/peer8.json 2> $O/peer8.err || { tail -5 $O/peer8.err; exit 1; }
timeout -k 10 200 python bench.py $A --force_peer --peer_shards 1 > $O/peer1.json 2> $O/peer1.err || { tail -5 $O/peer1.err; exit 1; }
timeout -k 10 300 python bench.py $A --force_partition > $O/rccl.json 2> $O/rccl.err || { tail -5 $O/rccl.err; exit 1; }
python - <<PY
import json
for k in ("single", "peer8", "peer1", "rccl"):
    d = json.loads(open("$O/%s.json" % k).read().strip().splitlines()[-1])
    c = d["config"]
    print(k, d["ms_per_step"], "ms/step;", c["graph_segments_per_step"], "segments,", c["collectives_per_step"], "collectives,",
          c["exchanged_MiB_per_step_per_gpu"], "MiB exchanged; status", d["status"])
PY
