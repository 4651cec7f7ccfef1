#!/bin/bash
O=gpurun_out/r2l; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -12 $O/bench.err; timeout 120 tools/bin/kexp5 batched > $O/kexp5_batched.log 2>&1; cat $O/kexp5_batched.log
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_all.log
