#!/bin/bash
# host profile of SVGDOptimizer.step + default bench + GPU suite (one box)
O=gpurun_out/${1:-check2}; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
python tools/shell_host_profile.py 2>&1 | grep -v amdgpu > $O/host_profile.txt; echo "host profile rc=$?"; grep "tensors" $O/host_profile.txt | cut -c1-200
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -i "shell_step\|swag_predict\|svgd_step:" $O/bench.err | cut -c1-400
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
