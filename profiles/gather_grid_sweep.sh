export GRAPES_DIAG=1   # (round 4) the A/B switches below exist in the diagnostic build only: libgrapes_hip_diag.so
cd $GRAFT_REPO_ROOT
for G in 768 1536 2304 3072 8192; do
  GRAPES_GATHER_GRID=$G timeout -k 5 200 python bench.py --steps 400 --warmup 300 --cpu_steps 0 > gpurun_out/sw_$G.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/sw_$G.json")); r=d["roofline"]["per_position"]
print($G, d["ms_per_step"], [ (p["us"], p["us_min"], p["us_max"]) for p in r[:4]])
PY
done
