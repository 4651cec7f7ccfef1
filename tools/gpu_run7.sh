#!/bin/bash
O=gpurun_out/r2m; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
for ex in alltoall pipelined allgather; do
  BDE_BENCH_DEVICE=0 BDE_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --blocks 2 --dim 4000000 --exchange $ex --no-extras > $O/bench_n2_$ex.json 2> $O/bench_n2_$ex.err; echo "bench n2 $ex rc=$?"; tail -1 $O/bench_n2_$ex.json | cut -c1-400
done
BDE_BENCH_DEVICE=0 BDE_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 4 --steps 2 --warmup 1 --blocks 1 --dim 1000000 --no-extras > $O/bench_n4.json 2> $O/bench_n4.err; echo "bench n4 rc=$?"; tail -1 $O/bench_n4.json | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -14 $O/bench.err | cut -c1-160
