# same-box A/B of one environment switch: bash profiles/ab_env.sh "<ENV=1 ...>" [rounds]
set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
for i in $(seq 1 ${2:-2}); do
for v in base "$1"; do
  if [ "$v" = base ]; then E=""; else E="$v"; fi
  env $E timeout -k 10 200 python bench.py $BENCH_ARGS --cpu_steps 0 --eager_steps 0 --no_roofline > gpurun_out/ab/line.json 2> gpurun_out/ab/err.log || { tail -5 gpurun_out/ab/err.log; exit 1; }
  python -c "import json; d=json.loads(open('gpurun_out/ab/line.json').read().strip().splitlines()[-1]); print('[$v]', d['ms_per_step'], d['ms_per_step_median'])"
done; done
