#!/usr/bin/env python3
"""End-to-end cost of SVGDOptimizer.step() minus a real model: 161 parameter tensors totalling
ResNet-50 size, trivial closures (loss = sum of <p, c>), so what is timed is the shell's host logic
(re-pointing views, zeroing rows, autograd hand-over) + the kernels + the base optimizer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beyond_deep_ensembles_amd as bde

dev = "cuda:0"
torch.manual_seed(0)
n_tensors, D = 161, 23_880_950
sizes = [D // n_tensors] * (n_tensors - 1)
sizes.append(D - sum(sizes))
M = 8


def run(fuse, reuse, base_kind, steps=10):
    params = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.05) for s in sizes]
    consts = [torch.randn(s, device=dev) * 0.01 for s in sizes]
    base = torch.optim.SGD(params, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=3e-4) if base_kind == "sgd" \
        else torch.optim.Adam(params, lr=1e-3)

    def reset():
        with torch.no_grad():
            for p in params[-2:]:
                p.normal_(0, 0.05)
    opt = bde.SVGDOptimizer(params, reset, base, particle_count=M, dataset_size=129809, fuse_base_optimizer=fuse, reuse_gram=reuse)
    fwd = lambda: sum(torch._foreach_mul(params, consts)[i].sum() for i in range(0, n_tensors, 40))  # touches a few tensors
    def fwd_all():
        prods = torch._foreach_mul(params, consts)
        return torch.stack([p.sum() for p in prods]).sum()
    for _ in range(2):
        opt.step(fwd_all, lambda l: l.backward())
    torch.cuda.synchronize()
    # closure cost alone
    t0 = time.perf_counter()
    for _ in range(steps):
        for i in range(M):
            for p in params: p.grad = None
            fwd_all().backward()
    torch.cuda.synchronize()
    t_closure = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.step(fwd_all, lambda l: l.backward())
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / steps
    print(f"fuse={fuse!s:5} reuse={reuse!s:5} base={base_kind:4}: step {t_step*1e3:8.2f} ms, of which M x fwd/bwd closures {t_closure*1e3:8.2f} ms "
          f"-> shell + kernels + optimizer {1e3*(t_step - t_closure):8.2f} ms", flush=True)


for args in [(False, False, "sgd"), (True, False, "sgd"), (True, True, "sgd"), (False, False, "adam"), (True, True, "adam")]:
    run(*args)
