#!/bin/bash
# round-2 GPU session 2
O=gpurun_out/r2b; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_dist_gpu.py > $O/pytest_main.log 2>&1; echo "pytest_main rc=$?"
timeout 600 python -m pytest tests/test_dist_gpu.py -q > $O/pytest_dist.log 2>&1; echo "pytest_dist rc=$?"
for sec in sample draw gram small batched; do timeout 300 tools/bin/kexp5 $sec; done > $O/kexp5.log 2>&1; echo "kexp5 rc=$?"
for v in timing base pnt ntnt; do timeout 300 tools/bin/kexp6_$v; done > $O/kexp6.log 2>&1; echo "kexp6 rc=$?"
BDE_SVGD_INPLACE_GRADS=1 BDE_NO_HOST_HELPER=1 timeout 600 python tools/shell_bench.py > $O/shell_bench_before.txt 2>&1; echo "shell before rc=$?"
timeout 600 python tools/shell_bench.py > $O/shell_bench_after.txt 2>&1; echo "shell after rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for ex in alltoall pipelined allgather; do
  BDE_BENCH_DEVICE=0 BDE_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --blocks 2 --dim 4000000 --exchange $ex --no-extras > $O/bench_n2_$ex.json 2> $O/bench_n2_$ex.err; echo "bench n2 $ex rc=$?"
done
tail -3 $O/pytest_main.log; tail -3 $O/pytest_dist.log; cat $O/shell_bench_before.txt $O/shell_bench_after.txt | tail -12; cat $O/bench_n2_*.json | cut -c1-600
