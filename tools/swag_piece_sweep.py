#!/usr/bin/env python3
"""Batched / single SWAG sampler and the moment update over row layouts: contiguous rows vs rows interleaved in pieces
of 2^lp floats, for the statistics and for the outputs independently (ResNet-50 size, K = 20, S = 30)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_deep_ensembles_amd import _lib
if len(sys.argv) > 1:                    # an alternative build of the library (e.g. tools/bin/libbde_norng.so)
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
theta = torch.randn(ld, device=dev, generator=g) * 0.05
out_flat = torch.empty(S, ld, device=dev)
o1 = torch.empty(ld, device=dev)
nb = 4 * D * (K + 2 + S)


def report(name, t, nbytes):
    print(f"{name:64s} {t*1e3:8.4f} ms {nbytes/t/1e9:8.1f} GB/s  {nbytes/t/8e12:6.3f}", flush=True)


report("batched: stats contiguous, out contiguous", bench.time_loop(lambda: ops.swag_sample_batched(mean, sq, ring, 3, out_flat, D, seed=1), 8), nb)
report("single : stats contiguous", bench.time_loop(lambda: ops.swag_sample(mean, sq, ring, 3, o1, D, seed=1, stream_id=2), 10), 4 * D * (K + 3))
report("update : stats contiguous", bench.time_loop(lambda: ops.swag_update(theta, mean, sq, ring[3], 5, D), 10), 24 * D)
for lp in (11, 12, 13):
    blk = RowBlock(K + 2, D, dev, log2_piece=lp)
    blk.buf.copy_(torch.randn(blk.buf.shape, device=dev, generator=g) * 1e-3)
    bm, bs, br = blk.row(K), blk.row(K + 1), blk.rows(0, K)
    ob = RowBlock(S, D, dev, log2_piece=lp)
    report(f"batched: stats pieces 2^{lp}, out contiguous",
           bench.time_loop(lambda: ops.swag_sample_batched(bm, bs, br, 3, out_flat, D, seed=1, pieces=blk.pieces), 8), nb)
    report(f"batched: stats contiguous, out pieces 2^{lp}",
           bench.time_loop(lambda: ops.swag_sample_batched(mean, sq, ring, 3, ob.rows(0, S), D, seed=1, out_pieces=ob.pieces), 8), nb)
    report(f"batched: stats pieces 2^{lp}, out pieces 2^{lp}",
           bench.time_loop(lambda: ops.swag_sample_batched(bm, bs, br, 3, ob.rows(0, S), D, seed=1, pieces=blk.pieces, out_pieces=ob.pieces), 8), nb)
    report(f"single : stats pieces 2^{lp}", bench.time_loop(lambda: ops.swag_sample(bm, bs, br, 3, o1, D, seed=1, stream_id=2, pieces=blk.pieces), 10), 4 * D * (K + 3))
    report(f"update : stats pieces 2^{lp}", bench.time_loop(lambda: ops.swag_update(theta, bm, bs, blk.row(3), 5, D, pieces=blk.pieces), 10), 24 * D)
    del blk, ob, bm, bs, br
    torch.cuda.empty_cache()
