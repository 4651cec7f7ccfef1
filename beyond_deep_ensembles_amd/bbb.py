"""Bayes by Backprop behind the reference's BBBOptimizer API.

Reference: ``src/algos/bbb.py`` -- ``GaussianPrior`` (:9-21), ``MixturePrior``
(:23-37) and ``BBBOptimizer`` (:43-99).  Same constructor, loss formula, NaN
guard and return value.  What changes: the means and rhos of all Gaussian
parameters are gathered ONCE into two flat device buffers (the parameters
become views), and the per-step KL collection loop (bbb.py:70-76: one
closed-form KL expression per tensor, differentiated again by autograd in
backward) is ONE fused kernel that returns the KL value and writes
``pi * dKL/dmean`` and ``pi * dKL/drho`` straight into the flat gradient
buffers before ``backward_closure`` accumulates the data-loss gradients on top.
Plain parameters get the same treatment for the ``l2_scale / 2 * ||p||^2`` term.
``MixturePrior`` has its own fused kernel (value + gradient wrt the means); tensor-valued priors and frozen
parameters keep the reference's autograd path.
"""
from __future__ import annotations

from typing import List

import torch

from .algo import BayesianOptimizer, FlatLayout, adopt_grads, check_params, clear_grads, _default_ops
from .util import GaussianParameter


class GaussianPrior:
    """N(mu, sigma^2) prior over every weight (API of ``src/algos/bbb.py:9-21``)."""

    def __init__(self, mu, sigma):
        self.mu, self.sigma = mu, sigma
        self.dist = torch.distributions.Normal(mu, sigma)

    def log_prob(self, x):
        return self.dist.log_prob(x)

    def kl_divergence(self, mu2, sigma2):
        """KL(N(mu2, sigma2^2) || prior), summed; the torch path that layers use.  BBBOptimizer takes the
        fused HIP kernel instead when the parameters come from GaussianParameter."""
        ratio = sigma2 / self.sigma
        shift = (self.mu - mu2) / self.sigma
        per_weight = 0.5 * (2 * torch.log(self.sigma / sigma2) - 1 + ratio.pow(2) + shift.pow(2))
        return per_weight.sum()


def _invalidate_sigma_caches() -> None:
    from .bbb_layers import invalidate_sigma_caches
    invalidate_sigma_caches()


class MixturePrior:
    """Scale mixture of two zero-mean Gaussians (API of ``src/algos/bbb.py:23-37``); its
    "KL" is the negative clamped log-density of the means, evaluated with torch autograd."""

    _LO, _HI = -23, 0

    def __init__(self, pi, sigma1, sigma2, validate_args=None):
        self.pi = torch.tensor(pi)
        self.sigma1, self.sigma2 = sigma1, sigma2
        self.dist1 = torch.distributions.Normal(0, sigma1, validate_args)
        self.dist2 = torch.distributions.Normal(0, sigma2, validate_args)

    def log_prob(self, value):
        parts = []
        for weight, dist in ((self.pi, self.dist1), (1 - self.pi, self.dist2)):
            parts.append(torch.log(weight) + torch.clamp(dist.log_prob(value), self._LO, self._HI))
        return torch.logaddexp(parts[0], parts[1])

    def kl_divergence(self, mu2, sigma2):
        return -self.log_prob(mu2).sum()


def collect_kl(model) -> torch.Tensor:
    """Sum of the ``kl`` attributes of all (nested) child layers."""
    total = 0
    for layer in model.children():
        total = total + getattr(layer, "kl", 0) + collect_kl(layer)
    return total


class _Group:
    """Flat storage of one param group: Gaussian (mean, rho) pairs and plain parameters."""

    def __init__(self, group, ops, device):
        self.prior = group["prior"]
        means, rhos, plain, generic = [], [], [], []
        claimed = set()
        params = group["params"]
        gauss_ok = isinstance(self.prior, GaussianPrior) and not torch.is_tensor(self.prior.mu) \
            and not torch.is_tensor(self.prior.sigma)
        mixture_ok = isinstance(self.prior, MixturePrior) and not torch.is_tensor(self.prior.sigma1) \
            and not torch.is_tensor(self.prior.sigma2) and 0.0 < float(self.prior.pi) < 1.0
        self.kind = "gauss" if gauss_ok else ("mixture" if mixture_ok else None)
        fused_ok = self.kind is not None
        frozen_plain = []
        for p in params:
            mod = getattr(p, "_bde_gaussian", None)
            # the fused kernel writes gradients for mean AND rho, so both must be trainable; a frozen half keeps the
            # pair on the reference's autograd path, where it simply receives no gradient
            if hasattr(p, "get_parameter_kl") and fused_ok and isinstance(mod, GaussianParameter) \
                    and any(mod.rho is q for q in params) and p.requires_grad and mod.rho.requires_grad:
                means.append(p)
                rhos.append(mod.rho)
                claimed.add(id(p))
                claimed.add(id(mod.rho))
        for p in params:
            if id(p) in claimed:
                continue
            if hasattr(p, "get_parameter_kl"):
                generic.append(p)                   # autograd path of the reference (bbb.py:73-74)
            elif not getattr(p, "_is_gaussian_mean", False) and not getattr(p, "_is_gaussian_rho", False):
                # l2 term (bbb.py:75-76); a frozen tensor contributes its value but never gets a gradient
                (plain if p.requires_grad else frozen_plain).append(p)
        self.means, self.rhos, self.plain, self.generic = means, rhos, plain, generic
        self.frozen_plain = frozen_plain
        self.gl = FlatLayout(means) if means else None
        self.pl = FlatLayout(plain) if plain else None

        def flatten(layout, plist):
            buf = torch.zeros(layout.ld, dtype=torch.float32, device=device)
            views = layout.views(buf)
            with torch.no_grad():
                torch._foreach_copy_(views, [p.detach() for p in plist])
            for p, v in zip(plist, views):
                p.data = v
            gbuf = torch.zeros(layout.ld, dtype=torch.float32, device=device)
            return buf, gbuf, layout.views(gbuf)

        if means:
            self.mu, self.gmu, self.gmu_views = flatten(self.gl, means)
            self.rho, self.grho, self.grho_views = flatten(self.gl, rhos)
            # GaussianParameter.sample() (rng="philox") draws the whole group in one launch through this object
            self._draw, self._consumed = None, []
            self._draw_grad_mode, self._draw_versions = True, []
            for i, p in enumerate(means):
                p._bde_gaussian._flat_group = self
                p._bde_gaussian._flat_index = i
        if plain:
            self.p, self.gp, self.gp_views = flatten(self.pl, plain)

    def invalidate_draw(self):
        """Forget the current group-wide draw (BBBOptimizer.step calls this when it starts and when it ends: the
        weights change there, and a draw never outlives the step it was made in)."""
        if self.means:
            self._draw, self._consumed = None, []

    def flat_sample(self, index, ops, seed):
        """Tensor ``index`` of the current group-wide draw (ONE launch for all tensors of the group).

        A draw is served only while it is known to be current: made in the same grad mode as this request, from the
        version of this tensor's mean / rho that is live now, and not yet handed out for this tensor.  Otherwise:

        * every tensor of the draw was consumed -> this is the first ``sample()`` of the next forward pass: new
          group-wide draw;
        * the draw is stale for this tensor (mean / rho modified since, other grad mode) -> new group-wide draw;
        * the tensor is asked again while others are still unconsumed (``mc_sample > 1`` inside one layer forward,
          or forward passes that touch different subsets of the group) -> an independent draw of THIS tensor only
          (``_GaussDraw``, 12 * numel bytes), the group-wide draw stays available to the tensors not yet served.
        """
        from .util import _FlatGaussDraw, _GaussDraw, _philox_stream
        mean, rho = self.means[index], self.rhos[index]
        grad_mode = torch.is_grad_enabled()
        current = self._draw is not None and self._draw_grad_mode == grad_mode \
            and self._draw_versions[index] == (mean._version, rho._version)
        if current and not self._consumed[index]:
            self._consumed[index] = True
            return self._draw[index]
        if current and not all(self._consumed):
            return _GaussDraw.apply(mean, rho, None, seed, next(_philox_stream), ops)
        self._draw = _FlatGaussDraw.apply(self, ops, seed, next(_philox_stream), *self.means, *self.rhos)
        self._draw_grad_mode = grad_mode
        self._draw_versions = [(m._version, r._version) for m, r in zip(self.means, self.rhos)]
        self._consumed = [False] * len(self.means)
        self._consumed[index] = True
        return self._draw[index]


class BBBOptimizer(BayesianOptimizer):
    '''
        Bayes By Backprop (drop-in for src/algos/bbb.py:43-99).  Use Bayesian layers built on
        GaussianParameter for the layers that should be treated as Bayesian.
    '''

    def __init__(self, params, base_optimizer, prior, dataset_size, mc_samples=1, kl_rescaling=1, components=1,
                 l2_scale=0, *, _ops=None):
        defaults = {"prior": prior, "l2_scale": l2_scale}
        super().__init__(params, defaults)
        self._ops = _ops or _default_ops()
        self.state["__base_optimizer"] = base_optimizer
        self._live_base = base_optimizer
        self.mc_samples = mc_samples
        self.kl_rescaling = kl_rescaling
        self.components = components
        self.dataset_size = dataset_size
        check_params(list(self._params()), self._ops)
        dev = self._params_device()
        self._groups: List[_Group] = [_Group(g, self._ops, dev) for g in self.param_groups]
        self._uncovered = None
        self._rws = self._ops.reduce_ws(dev)
        self._kl_parts = torch.zeros(2 * len(self._groups), dtype=torch.float32, device=dev)

    def _not_in_flat_buffers(self):
        """Every parameter base_optimizer.zero_grad() (bbb.py:60) would clear whose gradient the KL kernels do
        NOT overwrite: parameters on the autograd path (MixturePrior / tensor-valued priors / frozen halves, and
        their rho partners), frozen tensors, and parameters only the base optimizer knows."""
        if self._uncovered is None:
            covered = set()
            for fg in self._groups:
                covered.update(id(p) for p in fg.means + fg.rhos + fg.plain)
            seen, rest = set(), []
            base = self.state["__base_optimizer"]
            for group in list(self.param_groups) + list(base.param_groups):
                for p in group["params"]:
                    if id(p) not in covered and id(p) not in seen:
                        seen.add(id(p))
                        rest.append(p)
            self._uncovered = rest
        return self._uncovered

    def step(self, forward_closure, backward_closure, grad_scaler=None):
        # base_optimizer.zero_grad() (bbb.py:60): for the flat groups it is folded into the KL kernels below, which
        # OVERWRITE the flat gradient buffers with pi * dKL; every other parameter is cleared here (set to None, the
        # zero_grad default), so nothing accumulates from step to step.
        for p in self._not_in_flat_buffers():
            p.grad = None
        for fg in self._groups:
            fg.invalidate_draw()                        # a group-wide weight draw never outlives a step
        _invalidate_sigma_caches()                      # nor does a layer's cached sigma^2 (rho.data edits between steps)
        pi = self.kl_rescaling / self.dataset_size
        scale_dev = None
        if grad_scaler is not None and grad_scaler.is_enabled():
            if grad_scaler._scale is None:
                self.init_grad_scaler(grad_scaler)
            scale_dev = grad_scaler._scale              # backward_closure scales the loss by this tensor

        total_data_loss = None
        for _ in range(self.mc_samples):
            if total_data_loss is None:
                total_data_loss = forward_closure()
            else:
                total_data_loss += forward_closure()

        # collect KL loss & reg only once (bbb.py:69-76)
        total_kl_loss = torch.tensor(0.0, device=self._params_device())
        with torch.no_grad():
            for gi, (group, fg) in enumerate(zip(self.param_groups, self._groups)):
                if fg.means:
                    kl = self._kl_parts[2 * gi:2 * gi + 1]
                    if fg.kind == "gauss":
                        self._ops.gauss_kl(fg.mu, fg.rho, float(fg.prior.mu), float(fg.prior.sigma), fg.gl.d, self._rws,
                                           kl_out=kl, gmean=fg.gmu, grho=fg.grho, grad_scale=pi,
                                           grad_scale_dev=scale_dev, accumulate=False)
                    else:
                        # MixturePrior (bbb.py:31-37): a function of the means only; the rho gradients start at zero
                        self._ops.mixture_nll(fg.mu, float(fg.prior.pi), float(fg.prior.sigma1), float(fg.prior.sigma2),
                                              fg.gl.d, self._rws, val_out=kl, gmean=fg.gmu, grad_scale=pi,
                                              grad_scale_dev=scale_dev, accumulate=False)
                        fg.grho.zero_()
                    total_kl_loss = total_kl_loss + kl[0]
                    # the data-loss gradients of backward() come as fresh tensors and are added onto the KL
                    # gradients with one multi-tensor op (adopt_grads below), not one in-place add per tensor
                    clear_grads(fg.means)
                    clear_grads(fg.rhos)
                if fg.plain:
                    l2_scale = float(group["l2_scale"])
                    if l2_scale != 0.0:
                        val = self._kl_parts[2 * gi + 1:2 * gi + 2]
                        self._ops.l2(fg.p, l2_scale, fg.pl.d, self._rws, val_out=val, g=fg.gp, grad_scale=pi,
                                     grad_scale_dev=scale_dev, accumulate=False)
                        total_kl_loss = total_kl_loss + val[0]
                    else:
                        fg.gp.zero_()
                    clear_grads(fg.plain)
        # priors without a fused kernel keep the reference's autograd path
        for group, fg in zip(self.param_groups, self._groups):
            for p in fg.generic:
                total_kl_loss = total_kl_loss + p.get_parameter_kl(group["prior"])
            if group["l2_scale"] != 0:
                for p in fg.frozen_plain:
                    total_kl_loss = total_kl_loss + group["l2_scale"] / 2 * p.pow(2).sum()

        # don't divide the kl loss by the mc sample count as it has been collected only once (bbb.py:78-80)
        loss = pi * total_kl_loss + total_data_loss / (self.mc_samples * self.components)
        if not loss.isnan().any():
            backward_closure(loss)
            for fg in self._groups:
                if fg.means:
                    adopt_grads(fg.means, fg.gmu_views, add=True)
                    adopt_grads(fg.rhos, fg.grho_views, add=True)
                if fg.plain:
                    adopt_grads(fg.plain, fg.gp_views, add=True)

            if grad_scaler is not None:
                grad_scaler.step(self.state["__base_optimizer"])
            else:
                self.state["__base_optimizer"].step()
        for fg in self._groups:
            fg.invalidate_draw()
        _invalidate_sigma_caches()

        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._keep_live_base_optimizer(self._live_base)
        for fg in self._groups:
            fg.invalidate_draw()
        _invalidate_sigma_caches()

    def sample_parameters(self):
        '''The parameters sample themselves'''
        pass

    def get_base_optimizer(self):
        return self.state["__base_optimizer"]
