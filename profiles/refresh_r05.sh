# Round-5 evidence refresh (GPU box): bash profiles/refresh_r05.sh <tag> <commit> [skip_pmc]
# 1. python bench.py -> bench line   2. rocprofv3 --kernel-trace --stats of the same command   3. two --pmc passes (FETCH_SIZE,
# WRITE_SIZE; kernel-trace only, a dozen steps: counter collection serialises every dispatch) of the same command
# 4. profiles/make_bench_static.py -> bench_static.json   5. bench line again, whose frac uses the measured ramp + traffic.
# Copy gpurun_out/<tag>/{bench_line.json,kernel_stats.txt,bench_static.json} to profiles/.
set -o pipefail
TAG=${1:-r05}; COMMIT=${2:-unknown}; SKIP_PMC=${3:-0}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
O=gpurun_out/$TAG
( while sleep 50; do echo "[hb] $(date +%T) $(ls $O | wc -l) files"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 400 python bench.py > $O/bench_line0.json 2> $O/bench_err.log || { tail -5 $O/bench_err.log; exit 1; }
echo "[1] bench line"; python - <<PY
import json; d=json.load(open("$O/bench_line0.json")); r=d["roofline"] or {}; m=d.get("roofline_mfma") or {}
print("products", d["ms_per_step"], "ms/step median", d.get("ms_per_step_median"), d["value"], "roof", r.get("frac"), r.get("frac_all_positions_stamps"), r.get("frac_frontier_launches_stamps"), "mfma", m.get("frac"), m.get("fp32_equivalent_tflops"))
print("eager", (d["config"].get("eager_dropin_loop") or {}).get("ms_per_step"))
PY
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --steps 500 --warmup 500 --cpu_steps 0 --eager_steps 0 > $O/bench_prof.log 2>&1 || { tail -5 $O/bench_prof.log; exit 2; }
python profiles/summarize_rocprof.py $O/prof > $O/kernel_stats.txt
echo "[2] kernel stats"; head -12 $O/kernel_stats.txt
if [ "$SKIP_PMC" = "0" ]; then
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python bench.py --steps 8 --warmup 4 --cpu_steps 0 --eager_steps 0 --no_roofline --no_median > $O/pmc_f.log 2>&1 || { tail -5 $O/pmc_f.log; exit 3; }
echo "[3a] fetch pass done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python bench.py --steps 8 --warmup 4 --cpu_steps 0 --eager_steps 0 --no_roofline --no_median > $O/pmc_w.log 2>&1 || { tail -5 $O/pmc_w.log; exit 4; }
echo "[3b] write pass done"
fi
echo "[4] static"; python profiles/make_bench_static.py $O/bench_line0.json $O/prof $O/pmc_f $O/pmc_w $COMMIT $O/bench_static.json || exit 5
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
cp $O/bench_static.json profiles/bench_static.json
timeout -k 10 400 python bench.py > $O/bench_line.json 2>> $O/bench_err.log || { tail -5 $O/bench_err.log; exit 6; }
python - <<PY
import json; d=json.load(open("$O/bench_line.json")); r=d["roofline"]
print("[5] final", d["ms_per_step"], r["frac"], r["frac_all_positions_stamps"], r["frac_frontier_launches_stamps"], r["traffic"])
PY
echo done
