#!/bin/bash
O=gpurun_out/r4d; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv_lrt" > $O/pytest_conv.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_conv.log | cut -c1-200
for i in 1 2; do
timeout 300 python tools/swag_batched_ab.py >> $O/swag_batched_rounds_ab.txt 2>&1
timeout 300 python tools/swag_batched_ab.py tools/bin/libbde_philox10.so >> $O/swag_batched_rounds_ab.txt 2>&1
done
grep -v amdgpu.ids $O/swag_batched_rounds_ab.txt
