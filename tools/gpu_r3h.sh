#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
echo "=== Philox4x32-10 (product)"; timeout 600 python tools/swag_piece_sweep.py 2>&1 | grep -v amdgpu.ids | grep "batched\|single" | tee $O/sweep_p10.txt
echo "=== Philox4x32-7 (A/B build)"; timeout 600 python tools/swag_piece_sweep.py tools/bin/libbde_p7.so 2>&1 | grep -v amdgpu.ids | grep "batched\|single" | tee $O/sweep_p7.txt
echo "=== kernel trace of the real-gradient shell step"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o shell -- python3 $GRAFT_REPO_ROOT/tools/shell_host_profile.py > $GRAFT_REPO_ROOT/$O/shell_profile_under_rocprof.txt 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/shell_kernel_stats.csv
head -16 $O/shell_kernel_stats.csv | cut -c1-220
grep -v amdgpu $O/shell_profile_under_rocprof.txt | head -30
