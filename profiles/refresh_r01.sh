set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r01
timeout -k 10 500 python bench.py > gpurun_out/r01/bench_line.json 2> gpurun_out/r01/bench_err.log || exit 1
tail -c 600 gpurun_out/r01/bench_line.json
timeout -k 10 700 rocprofv3 --kernel-trace --stats -d gpurun_out/r01/prof -- python bench.py --cpu_steps 0 > gpurun_out/r01/bench_prof.log 2>&1 || exit 2
python profiles/summarize_rocprof.py gpurun_out/r01/prof > gpurun_out/r01/kernel_stats.txt
find gpurun_out/r01/prof -name "*.db" -delete; find gpurun_out/r01/prof -name "*.csv" -size +8M -delete
timeout -k 10 200 python profiles/split_gemm_check.py > gpurun_out/r01/split_gemm.txt 2>&1
timeout -k 10 200 python profiles/split_gemm_ablation.py >> gpurun_out/r01/split_gemm.txt 2>&1
echo done
