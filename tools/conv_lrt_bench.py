#!/usr/bin/env python3
"""bde_conv_lrt_fwd per ResNet-20 layer shape (batch 128) against the reference's op sequence in PyTorch on the same GPU
(two MIOpen convolutions + element-wise ops, bbb_layers.py:146-154) and against this package's round-3 composition
(stock convolutions + fused element-wise passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from beyond_deep_ensembles_amd.ops import HipOps
import bench

dev = torch.device("cuda", 0)
ops = HipOps()
shapes = [(128, 3, 32, 32, 16, 3, 1, 1), (128, 16, 32, 32, 16, 3, 1, 1), (128, 16, 32, 32, 32, 3, 2, 1), (128, 32, 16, 16, 32, 3, 1, 1),
          (128, 32, 16, 16, 64, 3, 2, 1), (128, 64, 8, 8, 64, 3, 1, 1), (128, 16, 32, 32, 32, 1, 2, 0),
          (32, 64, 56, 56, 64, 3, 1, 1), (32, 256, 14, 14, 256, 3, 1, 1), (32, 256, 56, 56, 64, 1, 1, 0)]
for n, c, h, w, o, k, s, p in shapes:
    x = torch.randn(n, c, h, w, device=dev)
    wm, wr = torch.randn(o, c, k, k, device=dev) * 0.1, torch.randn(o, c, k, k, device=dev) - 3.0
    bm, br = torch.randn(o, device=dev) * 0.1, torch.randn(o, device=dev) - 3.0
    ws2, bv = torch.empty_like(wm), torch.empty_like(bm)
    ops.var_operand_fwd(wr, 1, ws2)
    ops.var_operand_fwd(br, 2, bv)
    wbuf = ops.conv_lrt_wbuf(wm.shape, dev)
    ops.conv_lrt_prep(wm, wr, wbuf, br)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    out, var = torch.empty(n, o, ho, wo, device=dev), torch.empty(n, o, ho, wo, device=dev)
    if not ops.conv_lrt_supported(x.shape, wm.shape, (s, s), (p, p)):
        print((n, c, h, w, o, k, s, p), "unsupported")
        continue

    def fused():
        ops.conv_lrt_fwd(x, wbuf, wm.shape, bm, True, (s, s), (p, p), out, var, seed=1, stream_id=2)

    def torch_seq():
        mean = F.conv2d(x, wm, bm, stride=s, padding=p)
        v = F.conv2d((x ** 2).clamp(min=1e-4), (F.softplus(wr) ** 2).clamp(min=1e-4), F.softplus(br) ** 2, stride=s, padding=p)
        return mean + torch.sqrt(v) * torch.empty_like(mean).normal_(0, 1)

    def two_convs_only():
        F.conv2d(x, wm, bm, stride=s, padding=p)
        F.conv2d(x, ws2, bv, stride=s, padding=p)
    with torch.no_grad():
        tf, tt, tc = bench.time_loop(fused, 30), bench.time_loop(torch_seq, 30), bench.time_loop(two_convs_only, 30)
    flops = 2 * 2.0 * n * o * ho * wo * c * k * k
    print(f"N{n} C{c} {h}x{w} O{o} k{k} s{s}: fused {tf*1e6:8.1f} us ({flops/tf/1e12:5.1f} TFLOP/s)   torch sequence {tt*1e6:8.1f} us   "
          f"two MIOpen convs alone {tc*1e6:8.1f} us   speedup {tt/tf:5.2f}x", flush=True)
