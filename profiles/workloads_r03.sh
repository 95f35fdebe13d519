# Round-3 workload lines (GPU box): bash profiles/workloads_r03.sh -> gpurun_out/r03w/{reddit,arxiv,cora,products_random}.json + summary.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03w; O=gpurun_out/r03w; : > $O/summary.txt
for w in reddit arxiv cora; do
  timeout -k 10 300 python bench.py --workload $w --cpu_steps 8 --eager_steps 10 > $O/$w.json 2> $O/$w.err || { tail -3 $O/$w.err; exit 1; }
  python - <<PY >> $O/summary.txt
import json; d=json.load(open("$O/$w.json")); r=d.get("roofline") or {}; c=d.get("cpu_baseline") or {}
print("$w", d["ms_per_step"], "ms/step", d["value"], "edges/s | roofline", r.get("kernel"), r.get("frac"), "avg_launch_us", r.get("avg_launch_us"), "| cpu oracle ms/step", c.get("ms_per_step"), "| eager", (d["config"].get("eager_dropin_loop") or {}).get("ms_per_step"))
for p in r.get("per_position", []): print("    ", p)
PY
done
timeout -k 10 300 python bench.py --random_sampling --cpu_steps 8 --eager_steps 0 > $O/products_random.json 2> $O/products_random.err || exit 2
python -c "import json; d=json.load(open('$O/products_random.json')); print('products --random_sampling', d['ms_per_step'], 'ms/step', d['value'], 'edges/s | cpu oracle', (d.get('cpu_baseline') or {}).get('ms_per_step'))" >> $O/summary.txt
cat $O/summary.txt
