export GRAPES_DIAG=1   # (round 4) the A/B switches below exist in the diagnostic build only: libgrapes_hip_diag.so
set -e
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "shared_launch" 2>&1 | tail -3
python -m pytest tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
for v in 1 0; do
GRAPES_DW_DX_PAIR=$v python bench.py --cpu_steps 0 --eager_steps 0 --no_roofline > gpurun_out/pair_$v.$i.json
python - <<PY
import json; d=json.loads(open("gpurun_out/pair_$v.$i.json").read().strip().splitlines()[-1]); print("pair=$v", d["ms_per_step"], d["config"].get("graph_launches"))
PY
done; done
