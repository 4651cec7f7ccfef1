#!/bin/bash
O=gpurun_out/r4b; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_shells.py -m gpu -x -q -k "swag or streaming or rccl" > $O/pytest_swag.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_swag.log
timeout 600 python tools/swag_batched_ab.py > $O/swag_batched_ab.txt 2>&1; cat $O/swag_batched_ab.txt
timeout 900 python -m pytest tests/test_dist_gpu.py -m gpu -x -q -k "rccl" > $O/pytest_rccl.log 2>&1; echo "pytest rccl rc=$?"; tail -5 $O/pytest_rccl.log
bash tools/fault_hunt.sh 12 200 lazylog > $O/fault_hunt_lazylog.txt 2>&1; tail -60 $O/fault_hunt_lazylog.txt
bash tools/fault_hunt.sh 30 240 nokernels > $O/fault_hunt_nokernels.txt 2>&1; tail -8 $O/fault_hunt_nokernels.txt
