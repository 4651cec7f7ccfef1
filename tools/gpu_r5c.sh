#!/bin/bash
# third GPU call: AFTER tools/gpu_r5b.sh, with its table installed on the build side first
#   cp gpurun_out/r5b/conv_profit.json beyond_deep_ensembles_amd/conv_profit.json
# -- the layer now takes the fused kernels (with the pinned tilings) wherever the table says they win: the convolution tests and the
# CNN trajectory through the DEFAULT constructor on that table, then bench.py (its bbb_conv2d_* entries report `default_path`).
O=gpurun_out/r5c; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
python - <<'P' | tee $O/table.txt
import json
t = json.load(open("beyond_deep_ensembles_amd/conv_profit.json"))
print("abi", t.get("abi"), "source", t.get("source"))
for k, v in t.get("layers", {}).items():
    print(f"{k:40s} batch {v['batch']:4d} fwd {v['fwd']:5.2f}x fwd+bwd {v['fwd_bwd']:5.2f}x  fused {v['fused_us']:8.1f} us  reference {v['reference_us']:8.1f} us")
P
timeout 1200 python -m pytest tests -m gpu -q -k "conv or cnn" > $O/pytest_conv_with_table.log 2>&1; echo "conv tests rc=$?"; tail -15 $O/pytest_conv_with_table.log | cut -c1-220
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1200 $O/bench.err
