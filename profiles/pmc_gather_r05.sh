# Why does the step's gather-SpMM move 3.8 TB/s and not more?    Counter passes over `python bench.py` — kernel
# trace + --pmc only, one small set per pass — for gcn_aggregate_gather_head5_k: occupancy, where the wavefronts' cycles go (parked
# on a wait / issue stalls / executing), vector-memory latency, L2 hits and misses, reads that reach the fabric.
# usage (GPU box): bash profiles/pmc_gather_r05.sh   -> gpurun_out/pmc_gather/summary.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_gather
O=gpurun_out/pmc_gather
( while sleep 50; do echo "[hb] $(date +%T)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" \
           "MeanOccupancyPerCU GRBM_GUI_ACTIVE" \
           "VmemLatency" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python bench.py --steps 6 --warmup 4 --cpu_steps 0 --eager_steps 0 --no_roofline --no_median > $O/p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $O/p$i.log; }
done
python profiles/pmc_gather_parse.py $O > $O/summary.txt; cat $O/summary.txt
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
