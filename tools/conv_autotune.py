#!/usr/bin/env python3
"""Time EVERY candidate tiling of the fused BBBConv2d kernels on the device, pin the winners, and write the table
BBBConv2d(fused_conv="auto") reads (beyond_deep_ensembles_amd/conv_profit.py): per layer geometry the forward-only and
forward + backward speed-ups of the TUNED fused kernels over the reference's op sequence (bbb_layers.py:146-154 under
autograd) and the winning tilings, which `conv_profit.apply_tilings` pins at run time.

    python tools/conv_autotune.py [--out gpurun_out/conv_profit.json] [--iters 20] [--quick]

The planners' scores (csrc/conv_lrt.hip fwd_candidates, csrc/conv_lrt_bwd.hip wgrad_candidates) have hand-set weights that
were never compared with a device timing; this tool replaces the guess by a measurement: forward (one launch geometry),
input gradient over the zero-dilated gradient (one) and per phase (up to stride^2 launch geometries, tuned one at a time with
the others at their best so far), weight gradient (keyed by the layer).  Copy the output to
beyond_deep_ensembles_amd/conv_profit.json and this tool's log to profiles/.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import bench
from beyond_deep_ensembles_amd import conv_profit
from beyond_deep_ensembles_amd.ops import HipOps

LAYERS = [  # (N, C, H, W, O, K, stride, padding): the CIFAR ResNet-20 layers of BASELINE configs[1] at batch 128 ...
    (128, 3, 32, 32, 16, 3, 1, 1), (128, 16, 32, 32, 16, 3, 1, 1), (128, 16, 32, 32, 32, 3, 2, 1), (128, 32, 16, 16, 32, 3, 1, 1),
    (128, 32, 16, 16, 64, 3, 2, 1), (128, 64, 8, 8, 64, 3, 1, 1), (128, 16, 32, 32, 32, 1, 2, 0), (128, 32, 16, 16, 64, 1, 2, 0),
    # ... and ImageNet-sized ones (make_module_bbb over a ResNet-50, experiments/iwildcam/models.py:104-105)
    (32, 64, 56, 56, 64, 3, 1, 1), (32, 256, 14, 14, 256, 3, 1, 1), (32, 256, 56, 56, 64, 1, 1, 0), (32, 64, 56, 56, 256, 1, 1, 0)]


def main(argv=None, ops=None, dev=None, layers=None, time_loop=None, device_name=None):
    """(the keyword arguments exist for tests/test_hip_emu.py, which runs this tool's whole logic on the CPU model with a tiny layer)"""
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/conv_profit.json")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--quick", action="store_true", help="the first four layers only")
    a = ap.parse_args(argv)
    dev = torch.device("cuda", 0) if dev is None else dev
    ops = HipOps() if ops is None else ops
    time_loop = bench.time_loop if time_loop is None else time_loop
    layers = (LAYERS[:4] if a.quick else LAYERS) if layers is None else layers
    table = {"abi": int(ops.lib.bde_version()), "layers": {},
             "source": "tools/conv_autotune.py on " + (device_name or torch.cuda.get_device_name(0))}
    for n, c, h, w, o, k, s, p in layers:
        xs, wsh, st, pd = (n, c, h, w), (o, c, k, k), (s, s), (p, p)
        if not ops.conv_lrt_supported(xs, wsh, st, pd):
            print(xs, wsh, "unsupported", flush=True)
            continue
        x = torch.randn(*xs, device=dev)
        wm, wr = torch.randn(*wsh, device=dev) * 0.1, torch.randn(*wsh, device=dev) - 3.0
        bm, br = torch.randn(o, device=dev) * 0.1, torch.randn(o, device=dev) - 3.0
        wbuf = ops.conv_lrt_wbuf(wsh, dev)
        ops.conv_lrt_prep(wm, wr, wbuf, br, stride=st, padding=pd)
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        out, var = torch.empty(n, o, ho, wo, device=dev), torch.empty(n, o, ho, wo, device=dev)
        g = torch.randn_like(out)
        gvar, gx = torch.empty_like(g), torch.empty_like(x)
        gwm, gwr = torch.empty_like(wm), torch.empty_like(wr)

        def fwd():
            ops.conv_lrt_fwd(x, wbuf, wsh, bm, True, st, pd, out, var, seed=1, stream_id=2)

        def dgrad_dilated():
            ops.conv_lrt_bwd_data(g, gvar, wbuf, wsh, x, gx, st, pd)

        def dgrad_phases():
            ops.conv_lrt_bwd_data(g, gvar, wbuf, wsh, x, gx, st, pd, phases=True)

        def wgrad():
            ops.conv_lrt_bwd_weight(x, g, gvar, wr, gwm, gwr, st, pd)          # (sizes its partials buffer per call: follows the pin)

        gbm, gbr = torch.empty_like(bm), torch.empty_like(br)

        def gvar_pass():                                           # g_var + both bias gradients, one pass (what the layer's backward runs)
            ops.conv_lrt_gvar_bias(g, var, gvar, seed=1, stream_id=2, b_rho=br, g_bmu=gbm, g_brho=gbr)
        fwd()
        gvar_pass()

        def tune(which, fn):
            """Best time of the pass with every launch geometry pinned to its fastest candidate; [(geo, tiling, planner's time)]."""
            pinned = []
            for geo in ops.conv_lrt_pass_geos(which, xs, wsh, st, pd):
                cands, chosen = ops.conv_lrt_candidates(geo)
                times = []
                for cand in cands:
                    ops.conv_lrt_set_tiling(geo, cand)
                    times.append(time_loop(fn, a.iters))
                best = min(range(len(cands)), key=lambda i: times[i])
                ops.conv_lrt_set_tiling(geo, cands[best])
                pinned.append((geo, cands[best][:4]))
                print(f"    pass {which} geo {geo[5]}x{geo[6]} taps -> {geo[13]}x{geo[14]}: {len(cands)} tilings, planner's "
                      f"{cands[chosen][:4]} {times[chosen]*1e6:7.1f} us, best {cands[best][:4]} {times[best]*1e6:7.1f} us, worst "
                      f"{max(times)*1e6:7.1f} us", flush=True)
            return time_loop(fn, a.iters), pinned
        print(f"N{n} C{c} {h}x{w} O{o} k{k} s{s} p{p}", flush=True)
        t_f, pins_f = tune(0, fwd)
        t_dd, pins_dd = tune(1, dgrad_dilated)
        t_dp, pins_dp = (tune(2, dgrad_phases) if s > 1 else (t_dd, []))
        cands, chosen = ops.conv_lrt_wgrad_candidates(xs, wsh, st, pd)
        times = []
        for cand in cands:
            ops.conv_lrt_wgrad_set_tiling(xs, wsh, st, pd, cand)
            times.append(time_loop(wgrad, a.iters))
        best = min(range(len(cands)), key=lambda i: times[i])
        ops.conv_lrt_wgrad_set_tiling(xs, wsh, st, pd, cands[best])
        t_w = times[best]
        print(f"    weight gradient: {len(cands)} tilings, planner's {cands[chosen][:4]} {times[chosen]*1e6:7.1f} us, best "
              f"{cands[best][:4]} {t_w*1e6:7.1f} us, worst {max(times)*1e6:7.1f} us", flush=True)
        t_g = time_loop(gvar_pass, a.iters)

        # the reference's sequence on the same GPU: forward only, and forward + backward through autograd
        def torch_fwd():
            mean = F.conv2d(x, wm, bm, stride=s, padding=p)
            v = F.conv2d((x ** 2).clamp(min=1e-4), (F.softplus(wr) ** 2).clamp(min=1e-4), F.softplus(br) ** 2, stride=s, padding=p)
            return mean + torch.sqrt(v) * torch.empty_like(mean).normal_(0, 1)
        leaves = [t.clone().requires_grad_(True) for t in (x, wm, wr, bm, br)]
        noise = torch.randn_like(out)

        def torch_fwd_bwd():
            xx, m_, r_, bm_, br_ = leaves
            mean = F.conv2d(xx, m_, bm_, stride=s, padding=p)
            v = F.conv2d((xx ** 2).clamp(min=1e-4), (F.softplus(r_) ** 2).clamp(min=1e-4), F.softplus(br_) ** 2, stride=s, padding=p)
            torch.autograd.grad(mean + torch.sqrt(v) * noise, leaves, g)
        with torch.no_grad():
            t_tf = time_loop(torch_fwd, a.iters)
        t_tfb = time_loop(torch_fwd_bwd, a.iters)
        ours = t_f + t_g + min(t_dd, t_dp) + t_w
        flops = 2 * 2.0 * n * o * ho * wo * c * k * k
        print(f"    tuned: forward {t_f*1e6:7.1f} us ({flops/t_f/1e12:5.1f} TFLOP/s) vs torch {t_tf*1e6:7.1f} us = {t_tf/t_f:5.2f}x;  "
              f"forward + backward kernels {ours*1e6:7.1f} us (g_var {t_g*1e6:.1f}, input gradient dilated {t_dd*1e6:.1f} / per phase "
              f"{t_dp*1e6:.1f}, weight gradient {t_w*1e6:.1f}) vs torch autograd {t_tfb*1e6:7.1f} us = {t_tfb/ours:5.2f}x", flush=True)
        launch = pins_f + (pins_dp if (s > 1 and t_dp <= t_dd) else []) + pins_dd
        table["layers"][conv_profit._key(c, o, k, s, p, h, w)] = {
            "batch": n, "fwd": round(t_tf / t_f, 3), "fwd_bwd": round(t_tfb / ours, 3), "fused_us": round(ours * 1e6, 1),
            "reference_us": round(t_tfb * 1e6, 1), "forward_TFLOPs": round(flops / t_f / 1e12, 2),
            "tilings": {"launch": [list(geo) + list(til) for geo, til in launch], "wgrad": list(cands[best][:4])}}
        # leave no pins behind for the next layer's measurements of OTHER geometries (same-geometry phases share keys)
        for geo, _ in launch:
            ops.conv_lrt_set_tiling(geo, None)
        ops.conv_lrt_wgrad_set_tiling(xs, wsh, st, pd, None)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(table, f, indent=1)
    print("wrote", a.out)
    return table


if __name__ == "__main__":
    main()
