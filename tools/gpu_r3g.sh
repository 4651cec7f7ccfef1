#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
echo "=== batched sampler WITHOUT the noise epilogue (memory + MFMA ceiling of this kernel structure)"
timeout 600 python tools/swag_piece_sweep.py tools/bin/libbde_norng.so 2>&1 | grep -v amdgpu.ids | grep "batched\|library" | tee $O/sweep_norng.txt
echo "=== seg bench (next-chunk descriptor prefetched)"
timeout 600 python tools/seg_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/seg_bench.txt
echo "=== kernel trace of the real-gradient shell step"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o shell -- python3 $GRAFT_REPO_ROOT/tools/shell_host_profile.py > $GRAFT_REPO_ROOT/$O/shell_profile_under_rocprof.txt 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/shell_kernel_stats.csv
head -25 $O/shell_kernel_stats.csv | cut -c1-200
