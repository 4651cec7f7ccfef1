#!/bin/bash
O=gpurun_out/r4c; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv_lrt or swag" > $O/pytest_conv.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_conv.log
timeout 600 python tools/conv_lrt_bench.py > $O/conv_lrt_bench.txt 2>&1; cat $O/conv_lrt_bench.txt
bash tools/build_variant.sh philox10 "-DBDE_SWAG_PHILOX_ROUNDS=10" > /dev/null 2>&1
for i in 1 2; do
timeout 300 python tools/swag_batched_ab.py >> $O/swag_batched_rounds_ab.txt 2>&1
timeout 300 python tools/swag_batched_ab.py tools/bin/libbde_philox10.so >> $O/swag_batched_rounds_ab.txt 2>&1
done
grep -v amdgpu.ids $O/swag_batched_rounds_ab.txt
