#!/usr/bin/env python3
"""The fused BBBConv2d kernels per ResNet-20 layer shape (batch 128) and a few ImageNet-sized ones, each against the reference's
op sequence in PyTorch on the same GPU: forward (bde_conv_lrt_fwd vs two MIOpen convolutions + element-wise ops,
bbb_layers.py:146-154), input gradient (one launch over the zero-dilated gradient vs one launch per phase for strided
layers), weight gradient, and the backward of the reference's sequence through autograd.

--table [PATH]: also write the profitability table BBBConv2d(fused_conv="auto") reads (beyond_deep_ensembles_amd/conv_profit.py;
default PATH gpurun_out/conv_profit.json -- copy it to beyond_deep_ensembles_amd/conv_profit.json and file this tool's output under
profiles/): per layer geometry the forward-only and forward + backward speed-ups over the reference's sequence."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from beyond_deep_ensembles_amd.ops import HipOps
import bench

dev = torch.device("cuda", 0)
ops = HipOps()
table_path = None
if "--table" in sys.argv:
    i = sys.argv.index("--table")
    table_path = sys.argv[i + 1] if i + 1 < len(sys.argv) and not sys.argv[i + 1].startswith("-") else "gpurun_out/conv_profit.json"
table = {"abi": int(ops.lib.bde_version()), "source": "tools/conv_lrt_bench.py on " + torch.cuda.get_device_name(0), "layers": {}}
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
    ops.conv_lrt_prep(wm, wr, wbuf, br, stride=(s, s), padding=(p, p))
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
    # ---- backward: g_var + input gradient (dilated / per phase) + weight gradient vs autograd of the reference's sequence
    g = torch.randn_like(out)
    gvar, gx = torch.empty_like(g), torch.empty_like(x)
    gwm, gwr = torch.empty_like(wm), torch.empty_like(wr)
    wws = [None]

    def bwd_data_dilated():
        ops.conv_lrt_bwd_data(g, gvar, wbuf, wm.shape, x, gx, (s, s), (p, p))

    def bwd_data_phases():
        ops.conv_lrt_bwd_data(g, gvar, wbuf, wm.shape, x, gx, (s, s), (p, p), phases=True)

    def bwd_weight():
        wws[0] = ops.conv_lrt_bwd_weight(x, g, gvar, wr, gwm, gwr, (s, s), (p, p), ws=wws[0])

    gbm, gbr = torch.empty_like(bm), torch.empty_like(br)

    def gvar_pass():                                           # g_var + both bias gradients, one pass (what the layer's backward runs)
        ops.conv_lrt_gvar_bias(g, var, gvar, seed=1, stream_id=2, b_rho=br, g_bmu=gbm, g_brho=gbr)
    leaves = [t.clone().requires_grad_(True) for t in (x, wm, wr, bm, br)]
    noise = torch.randn_like(out)

    def torch_fwd_bwd():
        xx, m_, r_, bm_, br_ = leaves
        mean = F.conv2d(xx, m_, bm_, stride=s, padding=p)
        v = F.conv2d((xx ** 2).clamp(min=1e-4), (F.softplus(r_) ** 2).clamp(min=1e-4), F.softplus(br_) ** 2, stride=s, padding=p)
        torch.autograd.grad(mean + torch.sqrt(v) * noise, leaves, g)
    tg, td, tw = bench.time_loop(gvar_pass, 30), bench.time_loop(bwd_data_dilated, 30), bench.time_loop(bwd_weight, 30)
    tp = bench.time_loop(bwd_data_phases, 30) if s > 1 else td
    tref = bench.time_loop(torch_fwd_bwd, 20)
    ours = tf + tg + min(td, tp) + tw
    print(f"    backward: g_var {tg*1e6:7.1f}  input gradient {td*1e6:8.1f}" + (f" (per phase {tp*1e6:8.1f})" if s > 1 else "") +
          f"  weight gradient {tw*1e6:8.1f} us;  forward + backward kernels {ours*1e6:8.1f} us vs the reference's sequence through autograd "
          f"{tref*1e6:8.1f} us = {tref/ours:5.2f}x", flush=True)
    from beyond_deep_ensembles_amd import conv_profit
    table["layers"][conv_profit._key(c, o, k, s, p, h, w)] = {"batch": n, "fwd": round(tt / tf, 3), "fwd_bwd": round(tref / ours, 3),
                                                               "fused_us": round(ours * 1e6, 1), "reference_us": round(tref * 1e6, 1)}
if table_path:
    os.makedirs(os.path.dirname(os.path.abspath(table_path)), exist_ok=True)
    with open(table_path, "w") as f:
        json.dump(table, f, indent=1)
    print("wrote", table_path)
