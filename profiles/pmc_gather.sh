# PMC traffic of the step's gather-SpMM: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; nothing else traced) + parse.  usage (GPU box): bash profiles/pmc_gather.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/f -- python profiles/spmm_traffic.py > gpurun_out/pmc/f.log 2>&1 || { tail -5 gpurun_out/pmc/f.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc/w -- python profiles/spmm_traffic.py > gpurun_out/pmc/w.log 2>&1 || { tail -5 gpurun_out/pmc/w.log; exit 2; }
python profiles/spmm_traffic.py --parse gpurun_out/pmc/f gpurun_out/pmc/w
find gpurun_out/pmc -name "*.db" -delete
