#!/bin/bash
O=gpurun_out/r3m; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
python tools/swag_single_ab.py product=beyond_deep_ensembles_amd/lib/libbde_hip.so runtime_pieces_nopieces_build=tools/bin/libbde_nopieces.so product_again=beyond_deep_ensembles_amd/lib/libbde_hip.so 2>&1 | grep -v amdgpu | tee $O/single_ab.txt
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_philox.py -m gpu -x -q -k "swag or philox or single_launch" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep -i "swag_\|resnet20\|shell_step_ms" $O/bench.err | cut -c1-200
