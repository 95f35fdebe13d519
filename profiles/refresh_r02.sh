# Round-2 evidence refresh (run on the GPU box): bash profiles/refresh_r02.sh <tag> [workloads]
set -o pipefail
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
timeout -k 10 400 python bench.py > gpurun_out/$TAG/bench_line.json 2> gpurun_out/$TAG/bench_err.log || { tail -5 gpurun_out/$TAG/bench_err.log; exit 1; }
python - <<PY
import json; d=json.load(open("gpurun_out/$TAG/bench_line.json")); r=d["roofline"] or {}
print("products", d["ms_per_step"], "ms/step median", d.get("ms_per_step_median"), d["value"], "roof", r.get("frac"), r.get("avg_launch_us"))
for p in r.get("per_position", []): print("   ", p)
for p in (d.get("roofline_mfma") or {}).get("per_position", []): print("   ", p)
PY
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/prof -- python bench.py --cpu_steps 0 > gpurun_out/$TAG/bench_prof.log 2>&1 || exit 2
python profiles/summarize_rocprof.py gpurun_out/$TAG/prof > gpurun_out/$TAG/kernel_stats.txt
find gpurun_out/$TAG/prof -name "*.db" -delete; find gpurun_out/$TAG/prof -name "*.csv" -size +4M -delete
head -40 gpurun_out/$TAG/kernel_stats.txt
echo done
