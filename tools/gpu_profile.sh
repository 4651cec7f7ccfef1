#!/bin/bash
# rocprofv3 evidence for the round: kernel traces of bench.py (exact default command and the main loop alone) with
# the JSON of the SAME runs, PMC passes (traffic, MFMA busy, VALU activity), summaries.  Run from the repo root on the GPU box.
R=${1:-r03}
O=gpurun_out/prof_$R; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
export PYTHONDONTWRITEBYTECODE=1
# (a) exactly the driver's command (under rocprofv3 bench.py measures everything in ONE process: being_profiled())
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -o t -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-live-traffic > $O/bench_full_under_rocprof.json 2> $O/bench_full.err; echo "trace_full rc=$?"
# (b) the timed loop alone: every launch of the three SVGD kernels in the CSV belongs to the headline
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_main -o t -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-live-traffic > $O/bench_main_under_rocprof.json 2> $O/bench_main.err; echo "trace_main rc=$?"
# (c) PMC passes over the bench with extras (every kernel at ResNet-50 size), counters only; --extras-in-process: no child
#     process under the counter-collecting profiler (its preloaded library has initialised the GPU: a spawn there is an exec)
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAVES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -o p -- python3 bench.py --gpus 1 --steps 3 --warmup 1 --blocks 1 --no-cpu-baseline --no-config-extras --extras-in-process --no-live-traffic > $O/pmc_$tag.json 2> $O/pmc_$tag.err; echo "pmc $tag rc=$?"
done
# (d) the wide BBBLinear kernels at 4096 x 4096, batch 64 (tools/lrt_bench.py): kernel trace + the same counter groups
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_lrt -o t -- python3 tools/lrt_bench.py 64x4096x4096 > $O/lrt_bench_under_rocprof.txt 2> $O/lrt_trace.err; echo "trace_lrt rc=$?"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $O/lrtpmc_$tag -o p -- python3 tools/lrt_bench.py 64x4096x4096 > /dev/null 2> $O/lrtpmc_$tag.err; echo "lrt pmc $tag rc=$?"
done
BDE_PMC_LRT_SHAPE=64x4096x4096 python3 tools/pmc_summary.py $O/pmc_lrt_summary.json $O/lrtpmc_*/ ; echo "lrt summary rc=$?"
for f in $(find $O/trace_lrt -name "*kernel_stats.csv"); do mkdir -p $O/keep; cp $f $O/keep/lrt_kernel_stats.csv; done
rm -rf $O/trace_lrt $O/lrtpmc_*/
echo "--- sizes before pruning"; du -sh $O/* | sort -h | tail -12; find $O -type f -size +1M | head -20
f=$(find $O/pmc_FETCH_SIZE -name "*counter_collection*" | head -1); echo "counter file: $f"; head -2 "$f" | cut -c1-600
python3 tools/pmc_summary.py $O/pmc_summary.json $O/pmc_*/ ; echo "summary rc=$?"
# keep the summaries: stats CSVs, the JSON lines, the PMC summary; drop raw traces / per-dispatch tables / databases
mkdir -p $O/keep
for t in trace_full trace_main; do for f in $(find $O/$t -name "*kernel_stats.csv"); do cp $f $O/keep/${t}_kernel_stats.csv; done; done
cp $O/*.json $O/keep/ 2>/dev/null
for e in $O/*.err; do tail -5 $e > $O/keep/$(basename $e).tail; done
rm -rf $O/trace_full $O/trace_main $O/pmc_*/ $O/*.err
ls -la $O/keep
# plain run for comparison (no profiler attached)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_plain.json 2> $O/bench_plain.err; echo "plain rc=$?"
find $O -name "*kernel_stats.csv" | head; ls $O
