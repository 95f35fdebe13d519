cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/tl
O=gpurun_out/tl
timeout -k 10 ${TL_TIMEOUT:-300} rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py $BENCH_ARGS --steps 300 --warmup 300 --cpu_steps 0 --eager_steps 0 --no_roofline --no_median > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 2; }
python profiles/step_timeline.py $O/prof > $O/timeline.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
cat $O/timeline.txt
