#!/bin/bash
# Round 3, run D: segmented-gradient kernels (tests), host profile of the shell step, bench with the new extras.
O=gpurun_out/r3d; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_dist_fullsize_gpu.py > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log | cut -c1-300
timeout 600 python tools/shell_host_profile.py > $O/shell_host_profile.txt 2>&1; cat $O/shell_host_profile.txt | cut -c1-200
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -i "skipped\|real_grads\|shell" $O/bench.err | cut -c1-600
