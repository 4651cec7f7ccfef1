#!/bin/bash
O=gpurun_out/r3n; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log | cut -c1-300; grep "gave up" $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -i "swag_\|resnet20\|shell_step\|svgd_step:" $O/bench.err | cut -c1-260
