#!/bin/bash
O=gpurun_out/r2f; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_dist_gpu.py > $O/pytest_main.log 2>&1; echo "pytest_main rc=$?"
for v in timing base gnt grev gntrev base gnt; do timeout 300 tools/bin/kexp6_$v; done > $O/kexp6.log 2>&1; echo "kexp6 rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/pytest_main.log; grep -A10 "timestamps rep 2" $O/kexp6.log; grep "^== \|step: gram\|single launch\|^gram\|combine in place\|fused sgd" $O/kexp6.log
