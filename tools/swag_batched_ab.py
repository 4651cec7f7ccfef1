#!/usr/bin/env python3
"""Round 4: the batched SWAG sampler timed interleaved with its same-shape probe, K = 20 (LDS-DMA pipelined kernel)
and K = 30 (register kernel); argv[1]: optional alternative library build (e.g. a 10-round Philox variant from
tools/build_variant.sh) -- run the two builds in alternating processes on one box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_deep_ensembles_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from beyond_deep_ensembles_amd.ops import HipOps
import bench

dev = torch.device("cuda", 0)
ops = HipOps()
print("library:", _lib.LIB_PATH, "philox rounds of the samplers:", ops.swag_philox_rounds, flush=True)
probes = bench.StreamProbes()
D, S = 23_880_950, 30
ld = bench.pad_ld(D)
g = torch.Generator(device=dev).manual_seed(1)
for K in (20, 30):
    stat = torch.randn(K + 2, ld, device=dev, generator=g) * 1e-3
    stat[K + 1] += 1e-3
    ring, mean, sq = stat[:K], stat[K], stat[K + 1]
    out = torch.empty(S, ld, device=dev)
    nb = 4 * D * (K + 2 + S)
    arms = {"kernel": lambda: ops.swag_sample_batched(mean, sq, ring, 3, out, D, seed=1)}
    if K == 20 and probes.lib is not None:
        arms["probe R22 W30"] = lambda: probes.run((22, 30, 0), D, rd=stat, wr=out, nt_store=True)
    if K == 30 and probes.lib is not None:
        arms["probe R32 W30"] = lambda: probes.run((32, 30, 0), D, rd=stat, wr=out, nt_store=True)
    for fn in arms.values():
        bench.time_loop(fn, 5)
    times = {k: [] for k in arms}
    for rnd in range(5):
        for k, fn in arms.items():
            times[k].append(bench.time_loop(fn, 8))
    for k, ts in times.items():
        best, med = min(ts), sorted(ts)[len(ts) // 2]
        print(f"K={K} {k:16s} min {best*1e3:7.4f} ms ({nb/best/8e12:5.3f})  median {med*1e3:7.4f} ms ({nb/med/8e12:5.3f})  all "
              + " ".join(f"{t*1e3:.3f}" for t in ts), flush=True)
    del stat, out
    torch.cuda.empty_cache()
