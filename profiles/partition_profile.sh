# rocprofv3 kernel trace of the partitioned step at world size 1 (RCCL group of one): bash profiles/partition_profile.sh <tag>
TAG=${1:-pw1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/prof -- python bench.py --force_partition --steps 300 --warmup 50 --cpu_steps 0 --no_roofline > gpurun_out/$TAG/prof.log 2>&1 || exit 2
python profiles/summarize_rocprof.py gpurun_out/$TAG/prof 352 > gpurun_out/$TAG/kernel_stats.txt
find gpurun_out/$TAG/prof -name "*.db" -delete; find gpurun_out/$TAG/prof -name "*.csv" -size +4M -delete
head -30 gpurun_out/$TAG/kernel_stats.txt
