"""CPU oracle for the Bayesian-optimizer posterior-update hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``beyond_deep_ensembles_amd/`` may
import this module: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker (or the
timed CPU baseline), never as the thing shipped.

What it is: a restatement, in this repository's own words, of the arithmetic
behind ``BayesianOptimizer.step()/complete_epoch()/sample_parameters()`` of
Feuermagier/Beyond_Deep_Ensembles (``src/algos``).  Every function cites the
reference lines it follows (paths relative to the reference checkout).  The
arithmetic of the reference lives in PyTorch (un-vendored third party; the
reference pins it only informally as "PyTorch 2.0", ``Readme.md:64`` /
``setup.sh:4``); the oracle therefore issues the same ATen op sequence on CPU
tensors, so that with ``dtype=torch.float32`` it is bit-comparable to the
reference run on CPU, and with ``dtype=torch.float64`` it is the
high-precision evaluation that tolerances are stated against.

Parity pin: the reference has no tests or golden vectors for this path
(SURVEY.md section 4).  The oracle is pinned against outputs of the reference
itself, generated in the build container by ``oracle/gen_golden.py`` (which
imports ``/root/reference``) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks oracle == fixtures everywhere and
oracle == imported reference wherever ``/root/reference`` exists.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------
# SVGD
# --------------------------------------------------------------------------


def svgd_sq_dists(particles: torch.Tensor) -> torch.Tensor:
    """``d2 [M, M]``: squared Euclidean distances between the rows, formed as the reference forms them
    (``src/algos/svgd.py:15``): the p=2 ``cdist`` (a square root of the summed squares) squared again."""
    return torch.cdist(particles, particles, p=2) ** 2


def svgd_bandwidth_from_sq_dists(d2: torch.Tensor) -> torch.Tensor:
    """Median heuristic (``src/algos/svgd.py:18``): ``h = sqrt(0.5 * median / ln(M + 1)) + 1e-8`` where the
    "median" is ``torch.quantile(d2, 0.5)`` over ALL M*M entries -- the M zeros of the diagonal included,
    linear interpolation between the two middle order statistics."""
    m = d2.shape[0]
    return torch.sqrt(0.5 * torch.quantile(d2, 0.5) / np.log(m + 1)) + 1e-8


def svgd_kernel_from_sq_dists(d2: torch.Tensor, h) -> torch.Tensor:
    """``K = exp(-d2 / (2 h^2))`` (``src/algos/svgd.py:21``)."""
    return torch.exp(-d2 / (2 * h ** 2))


def svgd_repulsion(kernel: torch.Tensor, particles: torch.Tensor, h) -> torch.Tensor:
    """``gradK_i = (sum_j K_ij) x_i - sum_j K_ij x_j``, then divided by ``h^2`` (``src/algos/svgd.py:23,31``).
    Acts column by column on ``particles``, so it can be evaluated on a column slice with the full-width ``K``."""
    out = kernel.sum(dim=1).unsqueeze(-1) * particles - torch.matmul(kernel, particles)
    out /= h ** 2
    return out


def svgd_rbf(particles: torch.Tensor, h_override=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``rbf(particles, h_override)`` of ``src/algos/svgd.py:14-32``: returns ``(K [M, M], gradK [M, D])``.
    Composition of the four pieces above, in the reference's order (distances, bandwidth unless overridden,
    kernel, repulsive gradient)."""
    d2 = svgd_sq_dists(particles)
    h = svgd_bandwidth_from_sq_dists(d2) if h_override is None else h_override
    kernel = svgd_kernel_from_sq_dists(d2, h)
    return kernel, svgd_repulsion(kernel, particles, h)


def svgd_bandwidth(particles: torch.Tensor) -> torch.Tensor:
    """The median-heuristic bandwidth ``h`` alone (``src/algos/svgd.py:15,18``)."""
    return svgd_bandwidth_from_sq_dists(svgd_sq_dists(particles))


def svgd_direction(kernel: torch.Tensor, repulsion: torch.Tensor, particles: torch.Tensor, grads: torch.Tensor,
                   l2_reg: float, kernel_grad_scale: float, dataset_size: float) -> torch.Tensor:
    """``phi`` from a given kernel matrix and repulsive gradient (``src/algos/svgd.py:86,89``):
    the Gaussian-prior term ``l2_reg / 2 * theta`` is added to the gradients (Q2), then
    ``phi = K @ (-grads) + kernel_grad_scale * gradK / dataset_size`` -- no 1/M (Q1).  ``grads`` is left alone
    (the reference modifies its own stacked copy)."""
    with_prior = grads.clone()
    with_prior += l2_reg / 2 * particles
    return torch.matmul(kernel, -with_prior) + kernel_grad_scale * repulsion / dataset_size


def svgd_phi(particles: torch.Tensor, grads: torch.Tensor, l2_reg: float,
             kernel_grad_scale: float, dataset_size: float) -> torch.Tensor:
    """The SVGD direction ``phi [M, D]`` of one step (``src/algos/svgd.py:86-89``).  The reference then hands
    row i of ``-phi`` to the base optimizer as the gradient of particle i (``svgd.py:92-103``)."""
    kernel, repulsion = svgd_rbf(particles)
    return svgd_direction(kernel, repulsion, particles, grads, l2_reg, kernel_grad_scale, dataset_size)


def svgd_phi_cols(particle_cols: torch.Tensor, grad_cols: torch.Tensor, d2: torch.Tensor, l2_reg: float,
                  kernel_grad_scale: float, dataset_size: float, h_override=None) -> torch.Tensor:
    """``phi`` restricted to a column slice, given the FULL-width squared distances ``d2 [M, M]``: what one rank
    of the dimension-sharded multi-GPU update computes.  Same formulas as above; only
    the distances come from outside (they are sums over all columns)."""
    h = svgd_bandwidth_from_sq_dists(d2) if h_override is None else h_override
    kernel = svgd_kernel_from_sq_dists(d2, h)
    return svgd_direction(kernel, svgd_repulsion(kernel, particle_cols, h), particle_cols, grad_cols, l2_reg,
                          kernel_grad_scale, dataset_size)


def svgd_apply_shared_optimizer(rows: List[List[torch.Tensor]], neg_phi_rows: List[List[torch.Tensor]],
                                model_params: Sequence[torch.nn.Parameter], base_optimizer) -> None:
    """Apply the base optimizer once per particle, in particle order, with ONE
    optimizer whose state is keyed on the model's Parameters and therefore
    shared by all particles (``src/algos/svgd.py:92-103``).

    ``rows[i]`` are particle i's per-tensor storages (updated in place through
    ``param.data`` aliasing, line 96), ``neg_phi_rows[i]`` the per-tensor
    ``-phi`` slices that become ``param.grad`` (line 95).
    """
    for part, g in zip(rows, neg_phi_rows):
        for mp, pp, gg in zip(model_params, part, g):
            mp.grad = gg.clone()
            mp.data = pp
        base_optimizer.step()


# --------------------------------------------------------------------------
# SWAG
# --------------------------------------------------------------------------


@dataclass
class SwagState:
    """The statistics ``SwagOptimizer`` keeps (``src/algos/swag.py:28-35``)."""
    mean: torch.Tensor            # [D]      "__mean"
    sq_weights: torch.Tensor      # [D]      "__sq_weights"
    deviations: torch.Tensor      # [D, K]   "__deviations" (newest column is -1)
    epoch: int = 0                # "__epoch"
    steps_since_swag_start: int = 0   # "__steps_since_swag_start"
    updates: int = 0              # "__updates"
    # iterate tag carried by each deviation column (test bookkeeping only):
    column_iterate: List[int] = field(default_factory=list)


def swag_init(theta0: torch.Tensor, deviation_samples: int) -> SwagState:
    """``src/algos/swag.py:32-34``: mean = theta0, sq = theta0^2, dev = 0 [D, K].
    The initial weights count as sample #1."""
    mean = theta0.detach().clone()
    return SwagState(mean=mean, sq_weights=mean ** 2,
                     deviations=torch.zeros((mean.shape[0], deviation_samples), dtype=mean.dtype),
                     column_iterate=[-1] * deviation_samples)


def swag_gate(state: SwagState, start_epoch: int, update_interval) -> bool:
    """The integer schedule of ``_swag_update`` (``src/algos/swag.py:91-95``):
    after ``epoch >= start_epoch`` every optimizer step increments
    ``steps_since_swag_start``; an update fires iff that counter is a multiple
    of ``floor(update_interval)`` (``swag.py:19``).  Returns True iff the
    update fires (and then ``updates`` has been incremented, line 97).
    """
    interval = math.floor(update_interval)
    if state.epoch >= start_epoch:
        state.steps_since_swag_start += 1
        if state.steps_since_swag_start % interval == 0:
            state.updates += 1
            return True
    return False


def swag_moment_update(state: SwagState, theta: torch.Tensor, iterate_tag: int = -1) -> None:
    """The running moments and the deviation matrix (``src/algos/swag.py:98-104``),
    to be called after ``swag_gate`` returned True (it uses ``state.updates``
    as ``n``):
      * ``mean = (n * mean + theta) / (n + 1)``  (line 101)
      * ``sq   = (n * sq + theta^2) / (n + 1)``  (line 102)
      * ``dev  = roll(dev, -1, dim=1); dev[:, -1] = theta - mean_new``  (103-104)
    """
    n = state.updates
    state.mean = (n * state.mean + theta) / (n + 1)
    state.sq_weights = (n * state.sq_weights + theta ** 2) / (n + 1)
    state.deviations = torch.roll(state.deviations, -1, 1)
    state.deviations[:, -1] = theta - state.mean
    state.column_iterate = state.column_iterate[1:] + [iterate_tag]


def swag_complete_epoch(state: SwagState) -> None:
    """``src/algos/swag.py:60-61``."""
    state.epoch += 1


def swag_diag_and_factor(mean, sq_weights, deviations) -> Tuple[torch.Tensor, torch.Tensor]:
    """``src/algos/swag.py:112-113``: ``diag = 0.5 * (relu(sq - mean^2) + 1e-6)``,
    ``W = dev / sqrt(2 (K - 1))`` (the 1e-6 sits inside the 0.5; K is always
    ``deviation_samples``, also with fewer filled columns)."""
    k = deviations.shape[1]
    diag = 0.5 * (torch.relu(sq_weights.float() - mean.float() ** 2) + 1e-6)
    cov_factor = deviations.float() / math.sqrt(2 * (k - 1))
    return diag, cov_factor


def swag_sample(mean, sq_weights, deviations, eps_w: torch.Tensor, eps_d: torch.Tensor) -> torch.Tensor:
    """One posterior sample given the noise (``src/algos/swag.py:57,112-114`` +
    ``torch.distributions.LowRankMultivariateNormal.rsample``):
    ``theta = mean + W @ eps_W + sqrt(diag) * eps_D``.  The reference draws
    ``eps_W [K]`` first and ``eps_D [D]`` second."""
    diag, cov_factor = swag_diag_and_factor(mean, sq_weights, deviations)
    return mean.float() + torch.matmul(cov_factor, eps_w.unsqueeze(-1)).squeeze(-1) + diag.sqrt() * eps_d


def swag_build_dist(mean, sq_weights, deviations):
    """``_update_param_dist`` as the reference builds it (``src/algos/swag.py:107-114``),
    including the capacitance/Cholesky work of the distribution's constructor.
    Used for the timed CPU baseline and to confirm ``swag_sample``."""
    diag, cov_factor = swag_diag_and_factor(mean, sq_weights, deviations)
    return torch.distributions.LowRankMultivariateNormal(mean.float(), cov_factor, diag)


def swag_draw_noise(k: int, d: int, generator: Optional[torch.Generator] = None,
                    dtype=torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
    """Draw (eps_W, eps_D) in the order and by the calls ``rsample`` uses
    (``_standard_normal`` = ``torch.empty(shape).normal_()``; eps_W then eps_D)."""
    eps_w = torch.empty(k, dtype=dtype).normal_(generator=generator)
    eps_d = torch.empty(d, dtype=dtype).normal_(generator=generator)
    return eps_w, eps_d


# --------------------------------------------------------------------------
# Bayes by Backprop: mean-field Gaussian parameter
# --------------------------------------------------------------------------


def gauss_std(rho: torch.Tensor) -> torch.Tensor:
    """``std = softplus(rho)``  (``src/algos/util.py:181-183``)."""
    return torch.nn.functional.softplus(rho)


def gauss_sample(mean: torch.Tensor, rho: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """``sample() = mean + eps * std``  (``src/algos/util.py:170-171``; eps is
    ``normal_like(std)``, ``util.py:185-186``)."""
    return mean + eps * gauss_std(rho)


def gauss_sample_backward(grad_out: torch.Tensor, rho: torch.Tensor, eps: torch.Tensor):
    """Analytic backward of ``gauss_sample`` (what autograd computes for
    ``util.py:170-171,183``): d/dmean = g, d/drho = g * eps * sigmoid(rho)."""
    return grad_out, grad_out * eps * torch.sigmoid(rho)


def gauss_kl(mean: torch.Tensor, rho: torch.Tensor, prior_mu: float, prior_sigma: float) -> torch.Tensor:
    """Closed-form KL(q || p) against a Gaussian prior, summed over elements
    (``src/algos/bbb.py:18-21`` called through ``util.py:173-174``):
    ``sum 0.5 * (2 ln(sp / s) - 1 + (s / sp)^2 + ((mp - m) / sp)^2)``."""
    sigma2 = gauss_std(rho)
    kl = 0.5 * (2 * torch.log(prior_sigma / sigma2) - 1 + (sigma2 / prior_sigma).pow(2)
                + ((prior_mu - mean) / prior_sigma).pow(2))
    return kl.sum()


def gauss_kl_grads(mean: torch.Tensor, rho: torch.Tensor, prior_mu: float, prior_sigma: float):
    """Analytic gradients of ``gauss_kl`` (what autograd yields for bbb.py:18-21):
    dKL/dmean = (m - mp) / sp^2 ; dKL/drho = (-1/s + s/sp^2) * sigmoid(rho)."""
    s = gauss_std(rho)
    g_mean = (mean - prior_mu) / (prior_sigma ** 2)
    g_rho = (-1.0 / s + s / (prior_sigma ** 2)) * torch.sigmoid(rho)
    return g_mean, g_rho


def mixture_nll(mean: torch.Tensor, pi: float, sigma1: float, sigma2: float) -> torch.Tensor:
    """``MixturePrior.kl_divergence`` (``src/algos/bbb.py:23-37``): minus the summed log-density of the means under
    ``pi N(0, sigma1) + (1 - pi) N(0, sigma2)``, each component's ``Normal.log_prob`` clamped to [-23, 0] before the
    ``logaddexp`` (lines 31-33).  Differentiable: autograd of this expression is the gradient oracle."""
    weight = torch.tensor(pi)
    parts = []
    for w, sigma in ((weight, sigma1), (1 - weight, sigma2)):
        log_density = torch.distributions.Normal(0, sigma).log_prob(mean)
        parts.append(torch.log(w) + torch.clamp(log_density, -23, 0))
    return -torch.logaddexp(parts[0], parts[1]).sum()


def mixture_nll_grad(mean: torch.Tensor, pi: float, sigma1: float, sigma2: float) -> torch.Tensor:
    with torch.enable_grad():
        m = mean.detach().clone().requires_grad_(True)
        mixture_nll(m, pi, sigma1, sigma2).backward()
    return m.grad


def bbb_loss(total_kl: torch.Tensor, total_data_loss: torch.Tensor, kl_rescaling: float,
             dataset_size: float, mc_samples: int, components: int) -> torch.Tensor:
    """``src/algos/bbb.py:78-80``: ``pi = kl_rescaling / dataset_size``;
    ``loss = pi * kl + data / (mc_samples * components)`` (the KL is collected
    once per step whatever ``mc_samples`` is)."""
    pi = kl_rescaling / dataset_size
    return pi * total_kl + total_data_loss / (mc_samples * components)


def l2_term(param: torch.Tensor, l2_scale: float) -> torch.Tensor:
    """``src/algos/bbb.py:75-76``: plain parameters add ``l2_scale / 2 * ||p||^2``."""
    return l2_scale / 2 * param.pow(2).sum()


# --------------------------------------------------------------------------
# iVON
# --------------------------------------------------------------------------


def ivon_init_precision(like: torch.Tensor, prior_prec: float, dataset_size: float) -> torch.Tensor:
    """``src/algos/ivorn.py:34``: precision starts at ``prior_prec / N`` (N
    without the augmentation factor)."""
    return torch.full_like(like, prior_prec / dataset_size)


def ivon_sample(mean: torch.Tensor, precision: torch.Tensor, n_eff: float, eps: torch.Tensor) -> torch.Tensor:
    """The weight-noise draw of ``sample_parameters`` (``src/algos/ivorn.py:102-111``):
    ``delta = 1 / sqrt(N * clamp(prec, 1e-4)) * eps`` with ``N = dataset_size *
    augmentation``; the parameter becomes ``mean + delta``.  Returns ``delta``."""
    return 1 / (n_eff * precision.clamp(min=1e-4)).sqrt() * eps


def ivon_update(mean, momentum, precision, delta_sum, acc_grad, *, step_t: int, lr: float,
                betas=(0.9, 0.999), prior_prec: float, dataset_size: float, damping: float = 0.0,
                tempering: float = 1.0, augmentation: float = 1.0, mc_samples: int = 1):
    """The natural-gradient update block (``src/algos/ivorn.py:66-89``).  ``step_t``
    is the already incremented step counter (line 69).  Returns the new
    ``(mean, momentum, precision)`` without touching the inputs.
    """
    n = dataset_size * augmentation                       # line 72
    delta = tempering * prior_prec / n                    # line 74
    return ivon_update_scalars(mean, momentum, precision, delta_sum, acc_grad, step_t=step_t, lr=lr, betas=betas,
                               lam=delta, n_eff=n, damping=damping, mc_samples=mc_samples)


def ivon_update_scalars(mean, momentum, precision, delta_sum, acc_grad, *, step_t: int, lr: float, betas,
                        lam: float, n_eff: float, damping: float, mc_samples: int):
    """Lines 79-89 of ``src/algos/ivorn.py`` given ``delta`` (= lam, line 74) and ``N`` (= n_eff, line 72)."""
    beta1, beta2 = betas
    t, n, delta = step_t, n_eff, lam
    gradient = acc_grad / mc_samples                      # line 79
    g_mu = delta * mean + gradient                        # line 80
    momentum = beta1 * momentum + (1 - beta1) * g_mu      # line 81
    g_s = delta - precision + (n * precision * delta_sum / mc_samples) * gradient + damping  # line 82
    corrected_momentum = momentum / (1 - beta1 ** t)      # line 84
    corrected_precision = precision / (1 - beta2 ** t)    # line 85
    mean = mean - lr * corrected_momentum / corrected_precision                      # line 88
    precision = precision + ((1 - beta2) + 0.5 * (1 - beta2) ** 2 * g_s / precision) * g_s  # line 89
    return mean, momentum, precision


# --------------------------------------------------------------------------
# DeepEnsemble ("MultiX") sample split
# --------------------------------------------------------------------------


def ensemble_split(samples: int, members: int) -> List[int]:
    """How ``DeepEnsemble.predict`` divides ``samples`` over its members
    (``src/algos/ensemble.py:37-40``): every member gets ``samples // members``
    and member 0 takes the remainder on top."""
    per = samples // members
    return [samples - (members - 1) * per] + [per] * (members - 1)


# --------------------------------------------------------------------------
# Timed CPU baselines (bench.py `cpu_baseline`, kind "port"): the same ATen op
# sequence the reference issues for the posterior update, on host cores.
# --------------------------------------------------------------------------


def cpu_svgd_step(particles: torch.Tensor, grads: torch.Tensor, l2_reg: float,
                  kernel_grad_scale: float, dataset_size: float) -> torch.Tensor:
    """``svgd.py:86-89`` on resident [M, D] tensors (no per-tensor gathers, no
    base-optimizer steps): the posterior-update block only."""
    return svgd_phi(particles, grads, l2_reg, kernel_grad_scale, dataset_size)


def cpu_swag_sample(mean, sq_weights, deviations) -> torch.Tensor:
    """One ``sample_parameters`` draw through the cached distribution
    (``swag.py:57``): ``LowRankMultivariateNormal.sample()``."""
    return swag_build_dist(mean, sq_weights, deviations).sample()
