#!/usr/bin/env python3
"""Kernel time of the segmented-gradient entry points against their flat-row twins at ResNet-50 size (161 tensors):
(a) segment pointers into ONE flat gradient buffer (isolates the chunk walk from memory placement),
(b) 8 x 161 separately allocated gradient tensors (what autograd hands over)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_deep_ensembles_amd.ops import HipOps
from beyond_deep_ensembles_amd.algo import FlatLayout
import bench

dev = torch.device("cuda", 0)
ops = HipOps()
M = 8


def run(n_tensors, D):
    sizes = [D // n_tensors] * (n_tensors - 1)
    sizes.append(D - sum(sizes))
    params = [torch.empty(s, device=dev) for s in sizes]
    lay = FlatLayout(params, align=4)
    d, ld = lay.d, lay.ld
    P = torch.randn(M, ld, device=dev) * 0.05
    G = torch.randn(M, ld, device=dev) * 0.01
    buf = torch.zeros(ld, device=dev)
    ws, ks = ops.svgd_ws(M, dev), ops.svgd_kstat(M, dev)
    ops.svgd_gram(P, d, ws)
    ops.svgd_kstats(ws, M, 0.0, 1.0, 129809.0, -1.0, ks)
    out = torch.empty_like(G)
    seg_flat = ops.seg_table(lay.offsets, lay.numels, M, dev)
    host = seg_flat.staging()
    for j in range(M):
        for s, v in enumerate(lay.views(G[j])):
            host[s * M + j] = v.data_ptr()
    seg_flat.upload()
    grads = [[torch.randn(s, device=dev) * 0.01 for s in sizes] for _ in range(M)]
    seg_sep = ops.seg_table(lay.offsets, lay.numels, M, dev)
    host = seg_sep.staging()
    for j in range(M):
        for s, g in enumerate(grads[j]):
            host[s * M + j] = g.data_ptr()
    seg_sep.upload()
    torch.cuda.synchronize()
    nb_c, nb_f = 12 * M * d, (12 * M + 8) * d
    rows = [
        ("combine flat", lambda: ops.svgd_combine(P, G, out, d, ks), nb_c),
        ("combine seg -> flat buffer", lambda: ops.svgd_combine_seg(P, seg_flat, out, d, ks), nb_c),
        ("combine seg -> separate tensors", lambda: ops.svgd_combine_seg(P, seg_sep, out, d, ks), nb_c),
        ("fused sgd flat", lambda: ops.svgd_fused_sgd(P, G, buf, d, ks, 1e-12, 0.9, 0.0, 3e-4, True, False), nb_f),
        ("fused sgd seg -> flat buffer", lambda: ops.svgd_fused_sgd_seg(P, seg_flat, buf, d, ks, 1e-12, 0.9, 0.0, 3e-4, True, False), nb_f),
        ("fused sgd seg -> separate tensors", lambda: ops.svgd_fused_sgd_seg(P, seg_sep, buf, d, ks, 1e-12, 0.9, 0.0, 3e-4, True, False), nb_f),
        ("fused sgd flat + next gram", lambda: ops.svgd_fused_sgd(P, G, buf, d, ks, 1e-12, 0.9, 0.0, 3e-4, True, False, ws_next=ws), nb_f),
        ("fused sgd seg + next gram -> separate", lambda: ops.svgd_fused_sgd_seg(P, seg_sep, buf, d, ks, 1e-12, 0.9, 0.0, 3e-4, True, False, ws_next=ws), nb_f),
        ("gather seg (separate -> flat rows)", lambda: ops.svgd_gather_seg(G, seg_sep, 0, M), 8 * M * d),
    ]
    print(f"--- {n_tensors} tensors, D = {D} (d padded {d}), {seg_sep.n_chunks} chunks")
    for name, fn, nb in rows:
        t = bench.time_loop(fn, 20)
        print(f"{name:40s} {t*1e3:8.4f} ms  {nb/t/1e9:8.1f} GB/s")


run(161, 23_880_950)
run(1, 23_880_950)
run(65, 273_610)
