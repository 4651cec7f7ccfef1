#!/bin/bash
# One GPU-box round trip: the default bench line + the GPU parity suite.  Usage (from the repo root):
#   gpurun --timeout 3000 -- 'bash tools/gpu_check.sh r2x'      -> gpurun_out/r2x/{bench.json,bench.err,pytest_all.log}
O=gpurun_out/${1:-check}; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -12 $O/bench.err
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_all.log
