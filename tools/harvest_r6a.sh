#!/bin/bash
# After tools/gpu_r6a.sh has run on the GPU box and gpurun has merged gpurun_out/r6a back: file the evidence under profiles/
# (the judged directory), install the verification records, show what to look at.  Run from the repo root, then review + commit.
set -u
O=gpurun_out/r6a
[ -d "$O" ] || { echo "no $O: the call has not run"; exit 1; }
for f in pytest_gpu_x.log pytest_gpu_full.log smoke.log device_verify.log conv_lrt_bench.txt; do
  [ -f "$O/$f" ] && { grep -v "amdgpu.ids" "$O/$f" | tail -c 60000 > "profiles/r06_${f%.*}.txt"; echo "filed profiles/r06_${f%.*}.txt"; }
done
for f in bench_plain.json bench_torchrun1.json; do
  [ -s "$O/$f" ] && { cp "$O/$f" "profiles/r06_$f"; echo "filed profiles/r06_$f"; }
done
[ -d "$O/verify" ] && for f in $O/verify/*.log; do tail -c 20000 "$f" > "profiles/r06_verify_$(basename ${f%.log}).txt"; done
# the rocprofv3 kernel traces of the same bench command (tools/gpu_r6_first.sh): stats CSVs + the JSON lines of the traced runs
P=gpurun_out/prof_r06/keep
if [ -d "$P" ]; then
  for f in $P/*; do cp "$f" "profiles/r06_$(basename $f)"; echo "filed profiles/r06_$(basename $f)"; done
  python - <<'P2'
import csv, glob
for f in glob.glob("profiles/r06_trace_main_kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        if "svgd_combine_kernel<8, true>" in row["Name"]:
            avg_ns = float(row["AverageNs"]); nb = 12 * 8 * 23_880_950
            print(f"roofline from the trace: svgd_combine_kernel<8, true> calls {row['Calls']} avg {avg_ns/1e3:.2f} us -> "
                  f"{nb/avg_ns:.1f} GB/s = {nb/avg_ns/8000:.3f} of 8 TB/s")
P2
fi
if [ -s gpurun_out/device_verified.json ]; then
  python - <<'P'
import json
new = json.load(open("gpurun_out/device_verified.json"))
from beyond_deep_ensembles_amd import device_verified as V
ok = {f: r for f, r in new.get("families", {}).items() if r.get("sha256") == V.source_hash(f)}
stale = sorted(set(new.get("families", {})) - set(ok))
if ok:
    table = V.load(V._PATH)
    table["families"].update(ok)
    table["note"] = "records written by tools/device_verify.py on the GPU box (tools/harvest_r6a.sh installed them)"
    json.dump(table, open(V._PATH, "w"), indent=1, sort_keys=True)
print("verified families installed:", sorted(ok), "| records for other sources ignored:", stale)
P
fi
echo "--- suite (-x):"; tail -3 "$O/pytest_gpu_x.log" 2>/dev/null
echo "--- full suite:"; tail -3 "$O/pytest_gpu_full.log" 2>/dev/null
echo "--- bench line:"; head -c 600 "$O/bench_plain.json" 2>/dev/null; echo
python -m beyond_deep_ensembles_amd.device_verified
echo "next: python tools/kernel_table.py (statuses by hand) --write; git add -A; git commit"
