#!/bin/bash
# Round 3, run J: the whole GPU suite on the current tree, default bench, lrt kernel times with and without the cache.
O=gpurun_out/r3j; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log | cut -c1-300; grep "gave up" $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -i "swag\|skipped\|real_grads\|shell_step\|svgd_step:" $O/bench.err | cut -c1-420
timeout 600 python tools/lrt_bench.py > $O/lrt_bench.txt 2>&1; grep -v amdgpu $O/lrt_bench.txt | tail -30
