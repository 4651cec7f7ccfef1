#!/bin/bash
# round 4, GPU call A: full GPU suite, bench under torchrun with ONE rank over RCCL, fault hunt
O=gpurun_out/r4a; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 \
  bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "bench rc=$?"
tail -c 1500 $O/bench_torchrun1.err
bash tools/fault_hunt.sh 60 ${HUNT_SEC:-700} lazy > $O/fault_hunt.txt 2>&1; tail -40 $O/fault_hunt.txt
