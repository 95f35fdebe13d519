export GRAPES_DIAG=1   # (round 4) the A/B switches below exist in the diagnostic build only: libgrapes_hip_diag.so
# The A/B switches of round 3 under the oracle tests (each switch alone; products / arxiv / reddit shapes): they must stay correct.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sw; : > gpurun_out/sw/summary.txt
for sw in GRAPES_HOP_COUNTED=0 GRAPES_HOP_CURSOR=1 GRAPES_GEMM_PAIR=0 GRAPES_DW_DX_PAIR=0 GRAPES_TSPLIT_DW_CW=4 GRAPES_COMPACT_WIDE=0 GRAPES_GATE_BITS=0 GRAPES_BWD_FORK=1 GRAPES_COMPACT_SMALL=0 GRAPES_HEAD_BWD_MULTI=0 GRAPES_IMAGES_ONE_LAUNCH=0 GRAPES_R1_BITS=0 GRAPES_FUSED_HEAD=0 GRAPES_HOP_COUNTED_MIN=65536; do
  env $sw timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "baseline_configs or variants" > gpurun_out/sw/$sw.txt 2>&1
  echo "$sw: $(tail -1 gpurun_out/sw/$sw.txt)" | tee -a gpurun_out/sw/summary.txt
done
