#!/bin/bash
# round-2 GPU session 1: full GPU test-suite, kernel experiments, bench
mkdir -p gpurun_out/r2a
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_dist_gpu.py > gpurun_out/r2a/pytest_main.log 2>&1; echo "pytest_main rc=$?" 
timeout 600 python -m pytest tests/test_dist_gpu.py -q > gpurun_out/r2a/pytest_dist.log 2>&1; echo "pytest_dist rc=$?"
timeout 600 tools/bin/kexp5 all > gpurun_out/r2a/kexp5.log 2>&1; echo "kexp5 rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; echo "bench rc=$?"
tail -5 gpurun_out/r2a/pytest_main.log; tail -15 gpurun_out/r2a/pytest_dist.log
