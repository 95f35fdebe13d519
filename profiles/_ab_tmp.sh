set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/reddit_prof -o r -- python bench.py --workload reddit --e_cap 524288 --cpu_steps 0 --no_roofline --steps 50 --warmup 5 > gpurun_out/reddit_prof.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/reddit_prof/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
