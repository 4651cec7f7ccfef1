#!/bin/bash
# round 6, second GPU call (after tools/gpu_r6a.sh is green): the round's rocprofv3 evidence on the final tree, the
# profitability table of the fused convolution, and the A/Bs queued since round 4.
O=gpurun_out/r6b; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
# every candidate tiling of every pass timed, the winners pinned, the tuned layers measured against the reference's sequence
timeout 1500 python tools/conv_autotune.py --out $O/conv_profit.json > $O/conv_autotune.txt 2>&1; grep -v amdgpu $O/conv_autotune.txt | tail -90
timeout 2400 bash tools/gpu_profile.sh r06 > $O/gpu_profile.log 2>&1; tail -30 $O/gpu_profile.log | cut -c1-200
timeout 300 python tools/graph_small_step_ab.py > $O/graph_small_step_ab.txt 2>&1; grep -v amdgpu $O/graph_small_step_ab.txt | tail -6
timeout 600 python tools/gram_split_ab.py > $O/gram_split_ab.txt 2>&1; grep -v amdgpu $O/gram_split_ab.txt | tail -30
bash tools/build_variant.sh philox10 "-DBDE_SWAG_PHILOX_ROUNDS=10" > $O/build_variant.log 2>&1   # (the variant must be of the current ABI)
for i in 1 2; do
timeout 300 python tools/swag_batched_ab.py >> $O/swag_batched_rounds_ab.txt 2>&1
timeout 300 python tools/swag_batched_ab.py tools/bin/libbde_philox10.so >> $O/swag_batched_rounds_ab.txt 2>&1
done
grep -v amdgpu.ids $O/swag_batched_rounds_ab.txt | tail -30
