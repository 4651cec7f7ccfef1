#!/bin/bash
O=gpurun_out/r2f; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_all.log 2>&1; echo "pytest rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
tail -25 $O/pytest_all.log; cat $O/smoke.log | tail -2
