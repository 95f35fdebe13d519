# Round-5 evidence, second half (GPU box): bash profiles/refresh_r05_extras.sh   -> gpurun_out/r05x/*
# step timelines (products, reddit), in-kernel phase stamps of the index chain, the other workloads' bench lines, the parity margins
# of the config tests, the evaluation benches.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05x
O=gpurun_out/r05x
( while sleep 50; do echo "[hb] $(date +%T) $(ls $O | wc -l) files"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
BENCH_ARGS="" TL_TIMEOUT=300 bash profiles/step_timeline.sh > /dev/null 2>&1; cp gpurun_out/tl/timeline.txt $O/step_timeline.txt; echo "[1] timeline products"; tail -1 $O/step_timeline.txt
BENCH_ARGS="--workload reddit" TL_TIMEOUT=300 bash profiles/step_timeline.sh > /dev/null 2>&1; cp gpurun_out/tl/timeline.txt $O/step_timeline_reddit.txt; echo "[2] timeline reddit"; tail -1 $O/step_timeline_reddit.txt
GRAPES_DIAG=1 GRAPES_LIB_PATH=grapes_amd/libgrapes_hip_stamps.so timeout -k 10 200 python profiles/index_phase_stamps.py > $O/index_phase_stamps.txt 2>&1; echo "[3] stamps rc=$?"
for w in reddit arxiv cora papers100m; do
  timeout -k 10 300 python bench.py --workload $w > $O/bench_$w.json 2>$O/err_$w.log || { tail -5 $O/err_$w.log; }
  python -c "
import json; d=json.load(open('$O/bench_$w.json')); print('[4] $w', d['ms_per_step'], d.get('ms_per_step_median'), (d['roofline'] or {}).get('frac'))"
done
rm -f $O/parity_margins.txt
GRAPES_PARITY_MARGINS_FILE=$O/parity_margins.txt timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -m gpu -q -k "baseline_configs or first_step_tolerances or partitioned_forms" > $O/pytest_margins.log 2>&1; echo "[5] margins rc=$?"; tail -2 $O/pytest_margins.log
timeout -k 10 300 python profiles/bench_eval.py --minibatch > $O/bench_eval_minibatch.json 2> $O/bench_eval_mb.err; echo "[6] eval minibatch rc=$?"
timeout -k 10 300 python profiles/bench_eval.py > $O/bench_eval_fullbatch.json 2> $O/bench_eval_fb.err; echo "[7] eval fullbatch rc=$?"
echo done
