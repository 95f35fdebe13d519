# BASELINE config 5 at its real shape on ONE MI355X (GPU box): bash profiles/run_papers100m.sh [tag] [steps]
# papers100M-like synthetic graph: N = 111,059,956, avg degree 29 symmetrised (~3.2e9 directed edges), F = 128, C = 172, 3 hops,
# classifier GCN(128,[256,256,172]); device CSR ingest (64-bit offsets) + captured training steps; keeps the bench line + HBM use.
TAG=${1:-r03p}; STEPS=${2:-200}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG; O=gpurun_out/$TAG
( while sleep 45; do echo "[hb] $(date +%T) $(rocm-smi --showmeminfo vram 2>/dev/null | grep -i used | head -1)"; done ) & HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 800 python bench.py --workload papers100m --steps $STEPS --warmup 50 --cpu_steps 0 --eager_steps 10 > $O/line.json 2> $O/err.log
RC=$?; echo rc=$RC; tail -5 $O/err.log; head -c 3000 $O/line.json; exit $RC
