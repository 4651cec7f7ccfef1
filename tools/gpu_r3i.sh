#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
echo "=== Philox4x32-10"; timeout 600 python tools/swag_layout_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/layout_ab_p10.txt
echo "=== Philox4x32-7";  timeout 600 python tools/swag_layout_ab.py tools/bin/libbde_p7.so 2>&1 | grep -v amdgpu.ids | tee $O/layout_ab_p7.txt
echo "=== Philox4x32-10 again"; timeout 600 python tools/swag_layout_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/layout_ab_p10_again.txt
