#!/usr/bin/env python3
"""the Gram pass loads the head of its walk non-temporally and the last `keep` bytes cacheably so
that the combine pass finds them in the Infinity Cache; `keep` = 240 MB was tuned at M = 8, D = 23.9 M only.  This
sweeps it at M = 5 (the reference's particle_count), 8 and 16: step time (gram + kstats + combine), interleaved rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_deep_ensembles_amd.ops import HipOps
import bench

dev = torch.device("cuda", 0)
ops = HipOps()
D = bench.D_RESNET50
ld = bench.pad_ld(D)
for M in (5, 8, 16):
    g = torch.Generator(device=dev).manual_seed(1)
    P = torch.zeros(M, ld, device=dev)
    P[:, :D] = torch.randn(D, device=dev, generator=g) * 0.05
    P[:, D - 372918:D] += (torch.rand(M, 372918, device=dev, generator=g) * 2 - 1) / 45.0
    G = torch.zeros(M, ld, device=dev)
    G[:, :D] = torch.randn(M, D, device=dev, generator=g) * 0.01
    out = torch.empty_like(G)
    ws, ks = ops.svgd_ws(M, dev), ops.svgd_kstat(M, dev)
    keeps = {"all nt (0)": 0, "120 MB": 120_000_000, "240 MB (default)": 240_000_000, "360 MB": 360_000_000,
             "all cacheable": 1 << 40}

    def step():
        ops.svgd_gram(P, D, ws)
        ops.svgd_kstats(ws, M, 0.0, 1.0, 129809.0, -1.0, ks)
        ops.svgd_combine(P, G, out, D, ks)
    times = {k: [] for k in keeps}
    for rnd in range(4):
        for name, keep in keeps.items():
            ops.svgd_set_gram_keep_bytes(keep)
            times[name].append(bench.time_loop(step, 12))
    ops.svgd_set_gram_keep_bytes(-1)
    for name, ts in times.items():
        print(f"M={M:2d} particle bytes {4*M*D/1e6:7.1f} MB  keep {name:18s} step min {min(ts)*1e3:.4f} ms  median "
              f"{sorted(ts)[len(ts)//2]*1e3:.4f} ms   all " + " ".join(f"{t*1e3:.4f}" for t in ts), flush=True)
    del P, G, out
    torch.cuda.empty_cache()
