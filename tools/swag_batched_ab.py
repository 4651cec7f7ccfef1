#!/usr/bin/env python3
"""Round 4 A/B of the batched SWAG sampler, interleaved and repeated in ONE process: register kernel (round 3) vs the
LDS-DMA pipelined kernel, each on contiguous rows and on rows in 16 KB pieces; beside them a same-shape stream probe is
not needed here (bench.py carries it).  BDE_BATCHED_KERNEL is read per call by this experimental build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_deep_ensembles_amd.ops import HipOps, RowBlock
import bench

dev = torch.device("cuda", 0)
ops = HipOps()
D, S = 23_880_950, 30
ld = bench.pad_ld(D)
g = torch.Generator(device=dev).manual_seed(1)
for K in (20, 30):
    mean = torch.randn(ld, device=dev, generator=g) * 0.05
    sq = mean * mean + 1e-4
    ring = torch.randn(K, ld, device=dev, generator=g) * 1e-3
    out_flat = torch.empty(S, ld, device=dev)
    nb = 4 * D * (K + 2 + S)
    blk = RowBlock(K + 2, D, dev)
    blk.buf.copy_(torch.randn(blk.buf.shape, device=dev, generator=g) * 1e-3)
    blk.buf[:, K + 1] += 1e-3
    ob = RowBlock(S, D, dev)

    def contiguous():
        ops.swag_sample_batched(mean, sq, ring, 3, out_flat, D, seed=1)

    def pieces():
        ops.swag_sample_batched(blk.row(K), blk.row(K + 1), blk.rows(0, K), 3, ob.rows(0, S), D, seed=1, pieces=blk.pieces,
                                out_pieces=ob.pieces)
    variants = {}
    for kern in ("0", "1"):
        for name, fn in (("contiguous", contiguous), ("pieces", pieces)):
            def run(fn=fn, kern=kern):
                os.environ["BDE_BATCHED_KERNEL"] = kern
                fn()
            variants[f"K={K} kernel {kern} ({'regs' if kern == '0' else 'lds-dma'}) {name}"] = run
    # same results from both kernels (bit for bit: same MFMA order, same epilogue arithmetic)
    os.environ["BDE_BATCHED_KERNEL"] = "0"; contiguous(); a = out_flat[:, :D].clone()
    os.environ["BDE_BATCHED_KERNEL"] = "1"; out_flat.zero_(); contiguous(); b = out_flat[:, :D]
    print(f"K={K}: kernels agree bit for bit: {bool(torch.equal(a, b))}  max abs diff {float((a - b).abs().max()):.3e}", flush=True)
    del a, b
    for fn in variants.values():
        bench.time_loop(fn, 5)
    times = {k: [] for k in variants}
    for rnd in range(5):
        for k, fn in variants.items():
            times[k].append(bench.time_loop(fn, 8))
    for k, ts in times.items():
        best, med = min(ts), sorted(ts)[len(ts) // 2]
        print(f"{k:44s} min {best*1e3:7.4f} ms ({nb/best/8e12:5.3f})  median {med*1e3:7.4f} ms ({nb/med/8e12:5.3f})  all "
              + " ".join(f"{t*1e3:.3f}" for t in ts), flush=True)
    del mean, sq, ring, out_flat, blk, ob
    torch.cuda.empty_cache()
