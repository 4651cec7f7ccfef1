#!/bin/bash
# round 6, first GPU call: the suite exactly as the driver runs it (pytest -x: verified kernels first, never-run code last),
# smoke(), bench.py as the driver runs it (plain, then one rank under torchrun), then the verification records
O=gpurun_out/r6a; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -x -q -m gpu --durations=20 > $O/pytest_gpu_x.log 2>&1; echo "suite (-x) rc=$?"; tail -40 $O/pytest_gpu_x.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_plain.json 2> $O/bench_plain.err; echo "bench plain rc=$?"; tail -c 1500 $O/bench_plain.err; head -c 2500 $O/bench_plain.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 \
  bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "bench torchrun rc=$?"; tail -c 1500 $O/bench_torchrun1.err; head -c 1500 $O/bench_torchrun1.json
# the device-unverified families, each in its own child pytest: green ones are recorded in gpurun_out/device_verified.json
timeout 2400 python tools/device_verify.py --out gpurun_out/device_verified.json --log-dir $O/verify > $O/device_verify.log 2>&1; echo "device_verify rc=$?"; tail -12 $O/device_verify.log
# whatever -x stopped short of: the whole suite without -x
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.log 2>&1; echo "full suite rc=$?"; tail -30 $O/pytest_gpu_full.log | cut -c1-220
timeout 600 python tools/conv_lrt_bench.py > $O/conv_lrt_bench.txt 2>&1; grep -v amdgpu $O/conv_lrt_bench.txt | tail -40
