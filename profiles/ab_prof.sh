# A/B helper (GPU box): bash profiles/ab_prof.sh <tag> "<ENV=1 ...>"  -> bench ms/step + rocprof kernel stats under gpurun_out/<tag>/
set -o pipefail
TAG=$1; ENVS=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
O=gpurun_out/$TAG
env $ENVS timeout -k 10 200 python bench.py $BENCH_ARGS --cpu_steps 0 --eager_steps 0 --no_roofline > $O/line.json 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
python -c "import json; d=json.load(open('$O/line.json')); print('$TAG', '$ENVS', d['ms_per_step'], d['ms_per_step_median'])"
if [ "${3:-1}" = "1" ]; then
for kv in $ENVS; do export $kv; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py $BENCH_ARGS --steps 300 --warmup 300 --cpu_steps 0 --eager_steps 0 --no_roofline --no_median > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 2; }
python profiles/summarize_rocprof.py $O/prof > $O/kernel_stats.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
head -34 $O/kernel_stats.txt | cut -c1-140
fi
