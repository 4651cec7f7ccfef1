#!/usr/bin/env python3
"""A/B of the batched SWAG sampler's row layouts, interleaved and repeated (the first timing of a process runs on a cold
device; single back-to-back comparisons on this pool are off by up to 20 %): contiguous rows vs rows in pieces of 2^lp
floats for statistics and outputs.  argv[1]: optional alternative library build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_deep_ensembles_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
    print("library:", _lib.LIB_PATH)
from beyond_deep_ensembles_amd.ops import HipOps, RowBlock
import bench

dev = torch.device("cuda", 0)
ops = HipOps()
D, K, S = 23_880_950, 20, 30
ld = bench.pad_ld(D)
g = torch.Generator(device=dev).manual_seed(1)
mean = torch.randn(ld, device=dev, generator=g) * 0.05
sq = mean * mean + 1e-4
ring = torch.randn(K, ld, device=dev, generator=g) * 1e-3
out_flat = torch.empty(S, ld, device=dev)
nb = 4 * D * (K + 2 + S)
variants = {"contiguous": lambda: ops.swag_sample_batched(mean, sq, ring, 3, out_flat, D, seed=1)}
keep = []
for lp in (11, 12, 13):
    blk = RowBlock(K + 2, D, dev, log2_piece=lp)
    blk.buf.copy_(torch.randn(blk.buf.shape, device=dev, generator=g) * 1e-3)
    ob = RowBlock(S, D, dev, log2_piece=lp)
    keep.append((blk, ob))
    variants[f"pieces 2^{lp}"] = (lambda blk=blk, ob=ob: ops.swag_sample_batched(
        blk.row(K), blk.row(K + 1), blk.rows(0, K), 3, ob.rows(0, S), D, seed=1, pieces=blk.pieces, out_pieces=ob.pieces))
    variants[f"out pieces 2^{lp}, stats contiguous"] = (lambda ob=ob: ops.swag_sample_batched(
        mean, sq, ring, 3, ob.rows(0, S), D, seed=1, out_pieces=ob.pieces))
for fn in variants.values():            # warm the device and every code path
    bench.time_loop(fn, 5)
times = {k: [] for k in variants}
for rnd in range(4):
    for k, fn in variants.items():
        times[k].append(bench.time_loop(fn, 8))
for k, ts in times.items():
    best, med = min(ts), sorted(ts)[len(ts) // 2]
    print(f"{k:40s} min {best*1e3:7.4f} ms ({nb/best/8e12:5.3f})  median {med*1e3:7.4f} ms ({nb/med/8e12:5.3f})  all "
          + " ".join(f"{t*1e3:.3f}" for t in ts), flush=True)
