#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/seg_bench.py > $O/seg_bench.txt 2>&1; cat $O/seg_bench.txt
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_dist_fullsize_gpu.py > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -i "swag\|skipped\|real_grads\|shell_step" $O/bench.err | cut -c1-500
