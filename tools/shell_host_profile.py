#!/usr/bin/env python3
"""Where the HOST time of SVGDOptimizer.step goes at 161 tensors / ResNet-50 size (and 364 / DenseNet-121) with real
gradients: every helper of the step wrapped by a wall-clock timer (host side only: the kernels run asynchronously)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beyond_deep_ensembles_amd as bde
from beyond_deep_ensembles_amd import svgd as S
import bench

dev = torch.device("cuda", 0)


def profile(n_tensors, d, fuse=True, steps=20, producer="python node"):
    sizes = [d // n_tensors] * (n_tensors - 1)
    sizes.append(d - sum(sizes))
    params = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.05) for s in sizes]
    cs = [torch.randn(s, device=dev) * 0.01 for s in sizes]
    base = torch.optim.SGD(params, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=3e-4)
    opt = bde.SVGDOptimizer(params, lambda: None, base, particle_count=8, dataset_size=129809.0, fuse_base_optimizer=fuse,
                            reuse_gram=fuse)
    acc = {}

    def wrap(mod, name):
        fn = getattr(mod, name)

        def timed(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        setattr(mod, name, timed)
        return fn
    saved = {n: wrap(S, n) for n in ("repoint", "clear_grads", "collect_grads")}
    for n in ("_posterior_update", "_begin_particle", "_end_particle", "_release_grads", "_take_segments",
              "_prepare_and_check_grads", "_set_grad_scaler_state", "_check_single_launch", "_local_particles"):
        fn = getattr(opt, n)

        def timed(*a, _fn=fn, _n=n, **k):
            t0 = time.perf_counter()
            try:
                return _fn(*a, **k)
            finally:
                acc[_n] = acc.get(_n, 0.0) + time.perf_counter() - t0
        setattr(opt, n, timed)
    t_f = [0.0]

    def fwd():
        t0 = time.perf_counter()
        if producer == "python node":     # ONE Python autograd node: its gradients carry Python wrapper objects
            out = bench._ManyGrads.apply(cs, *params)
        else:                             # per-tensor C++ nodes (mul, sum), like a model: no Python objects on the gradients
            out = torch.stack([t.sum() for t in torch._foreach_mul(params, cs)]).sum()
        t_f[0] += time.perf_counter() - t0
        return out

    def bwd(loss):
        t0 = time.perf_counter()
        loss.backward()
        t_f[0] += time.perf_counter() - t0
    for _ in range(3):
        opt.step(fwd, bwd)
    torch.cuda.synchronize()
    acc.clear()
    t_f[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.step(fwd, bwd)
    host = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    for n, fn in saved.items():
        setattr(S, n, fn)
    print(f"{n_tensors} tensors, D = {d}, fused = {fuse}, gradients from {producer}: step {wall*1e3:.3f} ms wall ({host*1e3:.3f} ms until the last "
          f"launch was issued); closures {t_f[0]/steps*1e3:.3f} ms host; step() itself (not in any row below, "
          f"_release_grads / _take_segments / _check_single_launch are inside _posterior_update) "
          f"{(host - t_f[0]/steps - sum(v for k, v in acc.items() if k in ('_posterior_update', '_begin_particle', '_end_particle', '_prepare_and_check_grads', '_set_grad_scaler_state', '_local_particles'))/steps)*1e3:.3f} ms")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"    {k:20s} {v/steps*1e3:8.3f} ms per step")
    del opt, params, cs


for fuse in (True, False):
    profile(161, 23_880_950, fuse)
profile(161, 23_880_950, True)
profile(161, 23_880_950, True, producer="C++ nodes")
profile(364, 6_955_906, True)
profile(65, 273_610, True)
