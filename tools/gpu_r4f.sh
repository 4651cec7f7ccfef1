#!/bin/bash
# round 4, second call after the pool reopens: A/Bs and profiles
O=gpurun_out/r4f; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/gram_split_ab.py > $O/gram_split_ab.txt 2>&1; grep -v amdgpu $O/gram_split_ab.txt
for i in 1 2; do
timeout 300 python tools/swag_batched_ab.py >> $O/swag_batched_rounds_ab.txt 2>&1
timeout 300 python tools/swag_batched_ab.py tools/bin/libbde_philox10.so >> $O/swag_batched_rounds_ab.txt 2>&1
done
grep -v amdgpu.ids $O/swag_batched_rounds_ab.txt
bash tools/fault_hunt.sh 12 240 lazylog > $O/fault_hunt_lazylog.txt 2>&1; tail -60 $O/fault_hunt_lazylog.txt | cut -c1-240
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-config-extras > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; head -30 $O/kernel_stats.csv | cut -c1-200
