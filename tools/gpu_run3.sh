#!/bin/bash
O=gpurun_out/r2d; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_dist_gpu.py > $O/pytest_main.log 2>&1; echo "pytest_main rc=$?"
timeout 600 python -m pytest tests/test_dist_gpu.py -q > $O/pytest_dist.log 2>&1; echo "pytest_dist rc=$?"
for v in timing base; do timeout 300 tools/bin/kexp6_$v; done > $O/kexp6.log 2>&1; echo "kexp6 rc=$?"
for sec in sample small; do timeout 300 tools/bin/kexp5 $sec; done > $O/kexp5.log 2>&1; echo "kexp5 rc=$?"
BDE_SVGD_INPLACE_GRADS=1 BDE_NO_HOST_HELPER=1 timeout 600 python tools/shell_bench.py > $O/shell_bench_before.txt 2>&1; echo "shell before rc=$?"
timeout 600 python tools/shell_bench.py > $O/shell_bench_after.txt 2>&1; echo "shell after rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/pytest_main.log; tail -3 $O/pytest_dist.log; cat $O/shell_bench_before.txt $O/shell_bench_after.txt; grep -A12 "timestamps rep 2" $O/kexp6.log; grep -B2 -A4 "single launch" $O/kexp6.log | head -30; head -8 $O/kexp5.log
