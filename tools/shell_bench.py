#!/usr/bin/env python3
"""Host-side cost of SVGDOptimizer.step(): 161 parameter tensors totalling ResNet-50 size, M = 8 particles.

Two closure sets, so that what is timed is the shell itself (re-pointing the views, gradient hand-over, kernel
launches, base optimizer), not a model:
  null     forward returns a constant, backward does nothing: step time = shell + kernels, nothing to subtract;
  flatdot  loss = <flat particle row, c> through ONE autograd node with 161 inputs whose backward hands out views
           of one fresh flat gradient (what a real model's backward produces: fresh, stealable tensors); the
           closures' own cost is measured separately and subtracted.

BDE_SVGD_INPLACE_GRADS=1 selects the round-1 gradient hand-over (param.grad pre-pointed at the flat row, autograd
accumulates in place: one add launch per tensor per backward) -- an A/B variant that lives HERE, as a subclass of
the optimizer, not in the product; BDE_NO_HOST_HELPER=1 runs the per-tensor loops in Python.  Output of both
settings is kept under profiles/."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beyond_deep_ensembles_amd as bde

from beyond_deep_ensembles_amd.algo import repoint


class InplaceGradsSVGD(bde.SVGDOptimizer):
    """Round-1 hand-over: the gradient row is zeroed and param.grad pointed at it, so backward() accumulates in place."""

    def _begin_particle(self, particle_idx):
        self._grad_row(particle_idx).zero_()
        repoint(self._plist, self._pviews[particle_idx], self._gviews[particle_idx])

    def _end_particle(self, particle_idx):
        pass


SVGD = InplaceGradsSVGD if os.environ.get("BDE_SVGD_INPLACE_GRADS") else bde.SVGDOptimizer
dev = "cuda:0"
torch.manual_seed(0)
n_tensors, D = 161, 23_880_950
sizes = [D // n_tensors] * (n_tensors - 1)
sizes.append(D - sum(sizes))
M = 8
print(f"settings: BDE_SVGD_INPLACE_GRADS={os.environ.get('BDE_SVGD_INPLACE_GRADS', '')!r} "
      f"BDE_NO_HOST_HELPER={os.environ.get('BDE_NO_HOST_HELPER', '')!r}; {n_tensors} tensors, D = {D}, M = {M}", flush=True)


class FlatDot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, row, c, sizes, *params):
        ctx.c, ctx.sizes = c, sizes
        return torch.dot(row, c)

    @staticmethod
    def backward(ctx, grad_out):
        fresh = ctx.c * grad_out                       # one kernel; its views are fresh tensors autograd can keep
        return (None, None, None) + tuple(torch.split(fresh, ctx.sizes))


def run(fuse, reuse, base_kind, closures, steps=10):
    params = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.05) for s in sizes]
    base = torch.optim.SGD(params, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=3e-4) if base_kind == "sgd" \
        else torch.optim.Adam(params, lr=1e-3)

    def reset():
        with torch.no_grad():
            for p in params[-2:]:
                p.normal_(0, 0.05)
    opt = SVGD(params, reset, base, particle_count=M, dataset_size=129809, fuse_base_optimizer=fuse, reuse_gram=reuse)
    c = torch.randn(D, device=dev) * 0.01
    zero = torch.zeros((), device=dev)
    ld = opt._layout.ld

    def current_row():
        idx = (params[0].data_ptr() - opt._P.data_ptr()) // (4 * ld)
        return opt._P[idx, :D]
    if closures == "null":
        fwd, bwd = (lambda: zero), (lambda l: None)
    else:
        fwd, bwd = (lambda: FlatDot.apply(current_row(), c, sizes, *params)), (lambda l: l.backward())
    for _ in range(2):
        opt.step(fwd, bwd)
    torch.cuda.synchronize()
    t_closure = 0.0
    if closures != "null":
        t0 = time.perf_counter()
        for _ in range(steps):
            for i in range(M):
                for p in params:
                    p.grad = None
                bwd(fwd())
        torch.cuda.synchronize()
        t_closure = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.step(fwd, bwd)
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / steps
    print(f"closures={closures:7} fuse={fuse!s:5} reuse={reuse!s:5} base={base_kind:4}: step {t_step*1e3:8.2f} ms, closures alone "
          f"{t_closure*1e3:7.2f} ms -> shell + kernels + optimizer {1e3*(t_step - t_closure):8.2f} ms", flush=True)
    del opt, params, base, c
    torch.cuda.empty_cache()


for closures in ("null", "flatdot"):
    for args in [(False, False, "sgd"), (True, False, "sgd"), (True, True, "sgd"), (False, False, "adam"), (True, True, "adam")]:
        run(*args, closures)
