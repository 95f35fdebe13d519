export GRAPES_DIAG=1   # (round 4) the A/B switches below exist in the diagnostic build only: libgrapes_hip_diag.so
# GPU box: bash profiles/fused_first_layer_probe.sh  -> gpurun_out/r03_fused/probe.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03_fused; O=gpurun_out/r03_fused/probe.txt; : > $O
make -C grapes_amd/csrc lb768 > gpurun_out/r03_fused/make.log 2>&1 || { tail -5 gpurun_out/r03_fused/make.log; exit 1; }
echo "# product library" >> $O; timeout -k 10 120 python profiles/fused_first_layer_probe.py >> $O 2>gpurun_out/r03_fused/err.log || exit 2
echo "# product library, gather restricted to 256 resident workgroups (4 wavefronts per CU)" >> $O; GRAPES_GATHER_GRID=256 timeout -k 10 120 python profiles/fused_first_layer_probe.py >> $O 2>>gpurun_out/r03_fused/err.log || exit 3
echo "# product library, gather restricted to 512 resident workgroups (8 wavefronts per CU)" >> $O; GRAPES_GATHER_GRID=512 timeout -k 10 120 python profiles/fused_first_layer_probe.py >> $O 2>>gpurun_out/r03_fused/err.log || exit 4
echo "# GEMM compiled with __launch_bounds__(768): 168 VGPRs, 82 spilled (hipcc -Rpass-analysis=kernel-resource-usage)" >> $O; GRAPES_LIB_PATH=$PWD/grapes_amd/libgrapes_hip_lb768.so timeout -k 10 120 python profiles/fused_first_layer_probe.py >> $O 2>>gpurun_out/r03_fused/err.log || exit 5
cat $O
