#!/bin/bash
O=gpurun_out/r2p; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -12 $O/bench.err; 
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_all.log
