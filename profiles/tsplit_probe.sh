cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/tp
rocprofv3 --kernel-trace --stats -d gpurun_out/tp/d0 -- python profiles/tsplit_probe.py > /dev/null 2>&1
python profiles/summarize_rocprof.py gpurun_out/tp/d0 | grep -E "tsplit|gemm_mfma"
rm -rf gpurun_out/tp
