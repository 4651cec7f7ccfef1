#!/bin/bash
# Round 3, run B: whole GPU suite with the bounded single launch, bench N = 1 with the in-run probe, first-launch A/B.
O=gpurun_out/r3b; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -x -q -s --deselect tests/test_dist_fullsize_gpu.py > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log | cut -c1-300
grep -h "gave up" $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -4 $O/bench.err | cut -c1-400
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3b/bench.json").read().strip().splitlines()[-1])
print(json.dumps(d["roofline"], indent=1))
print({k: d["extra"][k] for k in ("svgd_step_M8_resnet20",)})
PY
bash tools/stress_first_launch.sh
