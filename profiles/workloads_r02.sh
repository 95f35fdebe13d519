# rocprofv3 kernel trace of bench.py on the other BASELINE workloads (arxiv / reddit / cora): which kernels their steps run.
# usage (on the GPU box): bash profiles/workloads_r02.sh <tag>
set -o pipefail
TAG=${1:-r02a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
for WL in arxiv reddit cora; do
  EC=131072; [ $WL = reddit ] && EC=524288
  timeout -k 10 300 python bench.py --workload $WL --e_cap $EC --steps 300 --warmup 100 --cpu_steps 0 > gpurun_out/$TAG/bench_$WL.json 2> gpurun_out/$TAG/bench_$WL.err || exit 1
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/prof_$WL -- python bench.py --workload $WL --e_cap $EC --steps 300 --warmup 100 --cpu_steps 0 --no_roofline > gpurun_out/$TAG/prof_$WL.log 2>&1 || exit 2
  python profiles/summarize_rocprof.py gpurun_out/$TAG/prof_$WL 402 > gpurun_out/$TAG/kernel_stats_$WL.txt
  find gpurun_out/$TAG/prof_$WL -name "*.db" -delete; find gpurun_out/$TAG/prof_$WL -name "*.csv" -size +4M -delete
  python - <<PY
import json; d=json.load(open("gpurun_out/$TAG/bench_$WL.json")); print("$WL", d["ms_per_step"], "ms/step", d["value"], d["unit"])
PY
done
echo done
