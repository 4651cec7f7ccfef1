#!/usr/bin/env python3
"""HOST time of the optimizer shells' step() without a GPU: the product's HipOps over a stub library whose C-ABI entry
points return at once (generated from _lib.SIGNATURES, gcc), CPU tensors.  What is left is exactly the host work of a
step on the device: the Python of the shell, the C++ helper, argument checks, ctypes calls -- per step, with the closures'
own time subtracted.  (On the device the kernels run asynchronously beside this; bench.py's shell entries measure both.)

    python tools/shell_host_cpu.py [--profile]
"""
import argparse
import ctypes
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import beyond_deep_ensembles_amd as bde
from beyond_deep_ensembles_amd import _lib
from beyond_deep_ensembles_amd import ops as ops_mod

HERE = os.path.dirname(os.path.abspath(__file__))


def stub_library():
    out = os.path.join(HERE, "bin", "libbde_stub.so")
    src = os.path.join(HERE, "bin", "bde_stub.c")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    body = "#include <stddef.h>\n"
    for name, (res, _args) in _lib.SIGNATURES.items():
        if res is ctypes.c_char_p:
            body += f'const char* {name}() {{ return "stub"; }}\n'
        elif res is ctypes.c_size_t:
            body += f"size_t {name}() {{ return 1 << 20; }}\n"
        elif name.endswith("_supported"):
            body += f"int {name}() {{ return 1; }}\n"
        else:
            body += f"int {name}() {{ return 0; }}\n"
    if not os.path.exists(src) or open(src).read() != body:
        open(src, "w").write(body)
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-w", "-o", out, src])
    lib = ctypes.CDLL(out)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def stub_ops():
    ops = ops_mod.HipOps.__new__(ops_mod.HipOps)
    ops.lib = stub_library()
    ops.name = "hip_stub"
    ops.load_code_objects = lambda device: None
    ops_mod._ptr = lambda t, name="tensor": None if t is None else t.data_ptr()
    ops_mod._ptr64 = lambda t, name: t.data_ptr()
    ops_mod._stream = lambda: None
    return ops


class _ManyGrads(torch.autograd.Function):
    """One node that hands every parameter a fresh gradient tensor (what a model's backward does, without the model)."""
    @staticmethod
    def forward(ctx, cs, *params):
        ctx.cs = cs
        return params[0].new_zeros(())

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(c.clone() for c in ctx.cs)


def resnet20_shapes():
    """The 65 parameter tensors of the CIFAR ResNet-20 (BASELINE configs[1]) -- as MANY tensors, each shrunk to 8 elements:
    the host work of a step is per tensor, and on CPU tensors anything per element (clones, frees, torch's own math) would
    be real CPU time that the device does asynchronously."""
    n = 3 + 3 * 3 * 6 + 2 * 3 + 2                       # stem conv + bn, 9 blocks x (2 conv + 2 bn), 2 shortcuts, fc
    return [(8,)] * n


def measure(name, make, steps=300, profile=False):
    torch.manual_seed(0)
    shapes = resnet20_shapes()
    params = [torch.nn.Parameter(torch.randn(sh) * 0.05) for sh in shapes]
    cs = [torch.randn(sh) * 0.01 for sh in shapes]
    opt = make(params)
    t_cl = [0.0]

    def fwd():
        t0 = time.perf_counter()
        out = _ManyGrads.apply(cs, *params)
        t_cl[0] += time.perf_counter() - t0
        return out

    def bwd(loss):
        t0 = time.perf_counter()
        loss.backward()
        t_cl[0] += time.perf_counter() - t0
    for _ in range(20):
        opt.step(fwd, bwd)
    best = None
    for _ in range(5):
        t_cl[0] = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            opt.step(fwd, bwd)
        total = time.perf_counter() - t0
        host = (total - t_cl[0]) / steps * 1e6
        best = host if best is None else min(best, host)
    print(f"{name:46s} host {best:7.1f} us/step  (closures {t_cl[0] / steps * 1e6:6.1f} us/step, {len(params)} tensors)")
    if profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(steps):
            opt.step(fwd, bwd)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    ops = stub_ops()
    reset = lambda: None
    measure("SVGD 8 particles, Adam, default ctor (fused)", lambda p: bde.SVGDOptimizer(
        p, reset, torch.optim.Adam(p, lr=3e-5), particle_count=8, dataset_size=50000, _ops=ops), profile=a.profile)
    small = dict(single_launch="two", host_fast_paths=True)          # what the defaults become once device_verified.json holds records
    measure("SVGD 8 particles, Adam, small-model kernel + native host paths", lambda p: bde.SVGDOptimizer(
        p, reset, torch.optim.Adam(p, lr=3e-5), particle_count=8, dataset_size=50000, _ops=ops, **small))
    measure("SVGD 8 particles, nesterov SGD, default ctor (fused)", lambda p: bde.SVGDOptimizer(
        p, reset, torch.optim.SGD(p, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=3e-4), particle_count=8,
        dataset_size=50000, _ops=ops))
    measure("SVGD 8 particles, nesterov SGD, small-model kernel + native paths", lambda p: bde.SVGDOptimizer(
        p, reset, torch.optim.SGD(p, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=3e-4), particle_count=8,
        dataset_size=50000, _ops=ops, **small))
    measure("SVGD 8 particles, nesterov SGD, fused + reuse_gram", lambda p: bde.SVGDOptimizer(
        p, reset, torch.optim.SGD(p, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=3e-4), particle_count=8,
        dataset_size=50000, fuse_base_optimizer=True, reuse_gram=True, _ops=ops), profile=False)
    measure("SVGD 8 particles, Adam, fuse_base_optimizer=False", lambda p: bde.SVGDOptimizer(
        p, reset, torch.optim.Adam(p, lr=3e-5), particle_count=8, dataset_size=50000, fuse_base_optimizer=False, _ops=ops))
    measure("SWAG", lambda p: bde.SwagOptimizer(p, torch.optim.SGD(p, lr=1e-3, momentum=0.9), update_interval=1, start_epoch=0,
                                                deviation_samples=20, _ops=ops))
    measure("iVON, mc_samples=1", lambda p: bde.iVONOptimizer(p, lr=1e-2, prior_prec=50.0, dataset_size=50000, mc_samples=1, _ops=ops))


if __name__ == "__main__":
    main()
