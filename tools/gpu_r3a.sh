#!/bin/bash
# Round 3, run A: the full-size multi-rank parity tests + the N = 2 / N = 8 bench (all three exchange modes in one
# invocation) with every rank on the one device over gloo.
O=gpurun_out/r3a; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests/test_dist_fullsize_gpu.py -m gpu -x -q > $O/pytest_fullsize.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_fullsize.log
export BDE_BENCH_DEVICE=0 BDE_BENCH_BACKEND=gloo
for n in 2 8; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 3 --warmup 1 --blocks 2 > $O/bench_n${n}.json 2> $O/bench_n${n}.err; echo "bench n=$n rc=$?"
  tail -5 $O/bench_n${n}.err; head -c 3000 $O/bench_n${n}.json; echo
done
