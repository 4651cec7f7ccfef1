#!/bin/bash
# round 6, the call tools/gpu_poll.sh makes the moment the pool reopens: what the driver scores first (suite with -x, smoke(),
# bench.py plain and as one torchrun rank), then the rocprofv3 kernel traces of the same bench command (the roofline's second
# source), then the verification records of the never-run families, the suite without -x and the convolution table.
# Budget: argument 1 = seconds this call may take in total (default 5400); the later sections are skipped when it runs out.
BUDGET=${1:-5400}; T0=$(date +%s)
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
O=gpurun_out/r6a; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
SUITE_T=$(( BUDGET > 2700 ? 2400 : BUDGET - 300 )); [ $SUITE_T -lt 240 ] && SUITE_T=240
timeout $SUITE_T python -m pytest tests -x -q -m gpu --durations=20 > $O/pytest_gpu_x.log 2>&1; echo "suite (-x) rc=$?"; tail -40 $O/pytest_gpu_x.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_plain.json 2> $O/bench_plain.err; echo "bench plain rc=$?"; tail -c 1500 $O/bench_plain.err; head -c 2500 $O/bench_plain.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 \
  bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "bench torchrun rc=$?"; tail -c 1500 $O/bench_torchrun1.err; head -c 1500 $O/bench_torchrun1.json
echo "[left $(left) s]"
# kernel traces of the bench (program directly behind "--"; under the profiler bench.py measures everything in one process)
if [ $(left) -gt 900 ]; then
P=gpurun_out/prof_r06; mkdir -p $P/keep
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_main -o t -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-live-traffic > $P/keep/bench_main_under_rocprof.json 2> $P/bench_main.err; echo "trace_main rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_full -o t -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-live-traffic > $P/keep/bench_full_under_rocprof.json 2> $P/bench_full.err; echo "trace_full rc=$?"
for t in trace_full trace_main; do for f in $(find $P/$t -name "*kernel_stats.csv"); do cp $f $P/keep/${t}_kernel_stats.csv; done; done
for e in $P/*.err; do tail -5 $e > $P/keep/$(basename $e).tail; done
rm -rf $P/trace_full $P/trace_main $P/*.err
head -5 $P/keep/trace_main_kernel_stats.csv | cut -c1-200
fi
echo "[left $(left) s]"
# the device-unverified families, each in its own child pytest: green ones are recorded in gpurun_out/device_verified.json
if [ $(left) -gt 600 ]; then
timeout $(( $(left) - 120 )) python tools/device_verify.py --out gpurun_out/device_verified.json --log-dir $O/verify > $O/device_verify.log 2>&1; echo "device_verify rc=$?"; tail -12 $O/device_verify.log
fi
echo "[left $(left) s]"
# whatever -x stopped short of: the whole suite without -x
if [ $(left) -gt 900 ]; then
timeout $(( $(left) - 120 )) python -m pytest tests -q -m gpu > $O/pytest_gpu_full.log 2>&1; echo "full suite rc=$?"; tail -30 $O/pytest_gpu_full.log | cut -c1-220
fi
if [ $(left) -gt 400 ]; then
timeout 300 python tools/conv_lrt_bench.py > $O/conv_lrt_bench.txt 2>&1; grep -v amdgpu $O/conv_lrt_bench.txt | tail -40
fi
echo "[left $(left) s]"
