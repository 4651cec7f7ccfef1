"""Mean-field Gaussian parameter (Bayes by Backprop building block).

Reference: ``src/algos/util.py:151-202`` -- ``GaussianParameter`` (mean/rho
pair, ``std = softplus(rho)``, ``sample() = mean + eps * std``,
``kl_divergence(prior)``), ``normal_like`` and ``reset_model_params``.
``sample()`` runs the HIP draw kernel (forward) and its analytic backward
instead of ~4 ATen launches plus an autograd graph per tensor; the noise is
either drawn by torch (``rng="torch"``, the reference's stream) or generated
inside the kernel and re-generated in backward (``rng="philox"``: nothing is
saved for backward but the seed).
"""
from __future__ import annotations

import itertools
from typing import Callable, Optional

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable
import torch.nn.functional as F

_philox_stream = itertools.count()


def normal_like(tensor) -> torch.Tensor:
    """``torch.empty_like(tensor).normal_(0, 1)`` (util.py:185-186)."""
    return torch.empty_like(tensor).normal_(0, 1)


def non_mle_params(params):
    return filter(lambda p: getattr(p, "use_mle_training", False) is False, params)


def reset_model_params(model):
    '''Resets all parameters of the model that implement reset_parameters() (util.py:191-202)'''
    def weight_reset(m):
        reset_parameters = getattr(m, "reset_parameters", None)
        if callable(reset_parameters):
            m.reset_parameters()
    model.apply(weight_reset)


class _GaussDraw(torch.autograd.Function):
    """w = mean + softplus(rho) * eps with the HIP kernels (bde_gauss_draw_fwd/bwd)."""

    @staticmethod
    def forward(ctx, mean, rho, eps, seed, stream_id, ops):
        m, r = mean.detach().contiguous().view(-1), rho.detach().contiguous().view(-1)
        n = m.numel()
        w = torch.empty_like(m)
        e = None if eps is None else eps.contiguous().view(-1)
        ops.gauss_draw_fwd(m, r, w, n, eps=e, seed=seed, stream_id=stream_id)
        ctx.save_for_backward(r, e)
        ctx.meta = (seed, stream_id, ops, n, mean.shape)
        return w.view(mean.shape)

    @staticmethod
    @once_differentiable          # the kernels produce plain tensors: no double backward
    def backward(ctx, grad_out):
        r, e = ctx.saved_tensors
        seed, stream_id, ops, n, shape = ctx.meta
        g = grad_out.contiguous().view(-1)
        gmean, grho = torch.empty_like(g), torch.empty_like(g)
        ops.gauss_draw_bwd(g, r, gmean, grho, n, eps=e, seed=seed, stream_id=stream_id, accumulate=False)
        return gmean.view(shape), grho.view(shape), None, None, None, None


class _FlatGaussDraw(torch.autograd.Function):
    """ALL Gaussian parameters of a BBBOptimizer param group drawn by ONE launch over the group's flat
    mean / rho buffers (bde_gauss_draw_fwd, in-kernel Philox noise), and their gradients by one multi-tensor
    copy + ONE launch (bde_gauss_draw_bwd) -- instead of a launch pair per tensor (util.py:170-171 runs ~4 ATen
    launches per tensor per draw).  Inputs / outputs are the per-tensor views, so autograd sees ordinary tensors."""

    @staticmethod
    def forward(ctx, group, ops, seed, stream_id, *mean_and_rho):
        d, ld = group.gl.d, group.gl.ld
        w = torch.empty(ld, dtype=torch.float32, device=group.mu.device)
        ops.gauss_draw_fwd(group.mu, group.rho, w, d, eps=None, seed=seed, stream_id=stream_id)
        # backward re-reads the group's rho buffer (nothing is saved but the seed): remember which rho it was
        ctx.meta = (group, ops, seed, stream_id, sum(r._version for r in group.rhos))
        return tuple(group.gl.views(w))

    @staticmethod
    @once_differentiable          # the kernels produce plain tensors: no double backward
    def backward(ctx, *grads):
        group, ops, seed, stream_id, rho_versions = ctx.meta
        if sum(r._version for r in group.rhos) != rho_versions:
            raise RuntimeError("one of the rho parameters of this BBBOptimizer group was modified in place between the "
                               "forward pass that drew the weights and its backward pass; the draw's backward "
                               "re-reads rho (softplus'(rho) * eps) and would use the new values")
        d, ld = group.gl.d, group.gl.ld
        dev = group.mu.device
        g = torch.zeros(ld, dtype=torch.float32, device=dev) if any(x is None for x in grads) \
            else torch.empty(ld, dtype=torch.float32, device=dev)
        pairs = [(v, x) for v, x in zip(group.gl.views(g), grads) if x is not None]
        if pairs:
            torch._foreach_copy_([v for v, _ in pairs], [x for _, x in pairs])
        gmean = torch.empty(ld, dtype=torch.float32, device=dev)
        grho = torch.empty(ld, dtype=torch.float32, device=dev)
        ops.gauss_draw_bwd(g, group.rho, gmean, grho, d, eps=None, seed=seed, stream_id=stream_id, accumulate=False)
        return (None, None, None, None) + tuple(group.gl.views(gmean)) + tuple(group.gl.views(grho))


class GaussianParameter(nn.Module):
    '''
        A mean-field Gaussian parameter (drop-in for src/algos/util.py:151-183).
        Don't overwrite the rho *parameter*, or make sure to set _is_gaussian_rho on the new one.
    '''

    def __init__(self, size, device=None, *, rng="torch", seed=0, _ops=None):
        super().__init__()
        self.overwrite_mean(torch.empty(size, device=device))
        self.rho = nn.Parameter(torch.empty(size, device=device))
        self.rho._is_gaussian_rho = True   # required e.g. by BBB
        self.rng = rng
        self.seed = int(seed)
        self.noise_source: Optional[Callable[[torch.Tensor], torch.Tensor]] = None
        self._ops = _ops

    def blundell_init(self, mean_std=0.1):
        torch.nn.init.normal_(self.mean, 0, mean_std)
        torch.nn.init.constant_(self.rho, -3)

    def sign_init(self):
        with torch.no_grad():
            self.mean.data = (torch.rand_like(self.mean) > 0.5).float() * 2 - 1
        torch.nn.init.constant_(self.rho, -3)

    def _get_ops(self):
        if self._ops is None:
            from .algo import _default_ops
            self._ops = _default_ops()
        return self._ops

    def __getstate__(self):
        # the kernel backend wraps this process's handle of libbde_hip.so (ctypes): a pickled / deep-copied module takes
        # none along and binds the library again at its first use (torch.save(model) works as it does for the reference)
        state = dict(self.__dict__)
        from .ops import HipOps
        if isinstance(state.get("_ops"), HipOps):
            state["_ops"] = None
        return state

    def sample(self) -> torch.Tensor:
        # util.py:170-171: mean + normal_like(std) * std
        group = getattr(self, "_flat_group", None)
        if group is not None and self.rng == "philox" and self.noise_source is None:
            # owned by a BBBOptimizer: the whole group is drawn at once, this tensor takes its view
            return group.flat_sample(self._flat_index, self._get_ops(), self.seed)
        if self.noise_source is not None:
            eps = self.noise_source(self.rho)
        elif self.rng == "torch":
            eps = normal_like(self.rho)
        else:
            eps = None
        return _GaussDraw.apply(self.mean, self.rho, eps, self.seed, next(_philox_stream), self._get_ops())

    def kl_divergence(self, prior):
        return prior.kl_divergence(self.mean, self.std)

    def overwrite_mean(self, mean):
        self.mean = nn.Parameter(mean)
        self.mean.get_parameter_kl = self.kl_divergence
        self.mean._is_gaussian_mean = True   # required e.g. by BBB
        self.mean._bde_gaussian = self       # lets BBBOptimizer pair mean and rho for the fused KL kernel

    @property
    def std(self) -> torch.Tensor:
        return F.softplus(self.rho)
