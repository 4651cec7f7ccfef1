#!/usr/bin/env python3
"""HOST time of one fused BBBConv2d layer call (forward + backward) without a GPU: the product's layer over a stub library
whose C-ABI entry points return at once (tools/shell_host_cpu.py), Python autograd Function (the C++ node binds the device
library and cannot run here).  What is left is the interpreter + ctypes + planner cost per layer call -- the floor under the
fused path's kernels on the device (the C++ node removes most of the Python share there).  Caveat: the tensors are CPU
tensors, so torch's own element-wise work of the test harness (the `.sum()` of the output and its gradient) is real CPU time
here and asynchronous device time there: the forward-only figure (~50 us in the build container) is the cleaner one.

    python tools/conv_layer_host_cpu.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

import beyond_deep_ensembles_amd as bde
import beyond_deep_ensembles_amd.bbb_layers as BL
import shell_host_cpu as S


def main():
    ops = S.stub_ops()
    real = S._lib.load()                                   # the planners are host code: take them from the real library
    for name in ("bde_conv_lrt_supported", "bde_conv_lrt_prep_floats", "bde_conv_lrt_bwd_weight_ws_bytes", "bde_conv_lrt_gvar_ws_bytes",
                 "bde_version"):
        setattr(ops.lib, name, getattr(real, name))
    BL._native_nodes = lambda o: None
    prior = bde.GaussianPrior(0, 1.0)
    for n, c, hw, o, k, s, p in [(8, 16, 32, 16, 3, 1, 1), (8, 32, 16, 64, 3, 2, 1), (8, 64, 8, 64, 3, 1, 1)]:
        layer = bde.BBBConv2d(c, o, k, prior, prior, stride=s, padding=p, rng="philox", fused_conv=True, _ops=ops)
        x = torch.randn(n, c, hw, hw, requires_grad=True)
        leaves = [x, layer.weight.mean, layer.weight.rho, layer.bias.mean, layer.bias.rho]

        def fwd_bwd():
            torch.autograd.grad(layer(x).sum(), leaves)

        def fwd_only():
            with torch.no_grad():
                layer(x)
        for fn in (fwd_bwd, fwd_only):
            for _ in range(20):
                fn()
            best = None
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(200):
                    fn()
                t = (time.perf_counter() - t0) / 200 * 1e6
                best = t if best is None else min(best, t)
            print(f"C{c} {hw}x{hw} O{o} k{k} s{s}: {fn.__name__:9s} host {best:7.1f} us per call (Python Function, stub kernels)")


if __name__ == "__main__":
    main()
