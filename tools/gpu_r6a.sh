#!/bin/bash
# round 6, first GPU call: HEAD of round 5 on hardware -- new kernels first, then the whole suite (no -x), bench.py as the
# driver runs it (plain, then one rank under torchrun), conv bench last
O=gpurun_out/r6a; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "conv_lrt or swag_batched or small_model or r5_" > $O/pytest_new_kernels.log 2>&1; echo "new kernels rc=$?"; tail -30 $O/pytest_new_kernels.log | cut -c1-220
timeout 2400 python -m pytest tests -m gpu -q --durations=20 > $O/pytest_gpu_full.log 2>&1; echo "full suite rc=$?"; tail -60 $O/pytest_gpu_full.log | cut -c1-220
timeout 900 python bench.py > $O/bench_plain.json 2> $O/bench_plain.err; echo "bench plain rc=$?"; tail -c 1500 $O/bench_plain.err; head -c 2500 $O/bench_plain.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 \
  bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "bench rc=$?"; tail -c 1500 $O/bench_torchrun1.err; head -c 1500 $O/bench_torchrun1.json
timeout 600 python tools/conv_lrt_bench.py > $O/conv_lrt_bench.txt 2>&1; grep -v amdgpu $O/conv_lrt_bench.txt | tail -60
