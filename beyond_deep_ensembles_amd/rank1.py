"""Rank-1 variational layers: a deterministic weight matrix scaled by two sampled vectors.

The callers of ``GaussianParameter.sample()`` in the reference (``src/algos/rank1.py:9-131``; SURVEY.md
section 8f): ``y = W (x * s) * r + b`` with ``s`` (input side) and ``r`` (output side) drawn per forward
pass from mean-field Gaussians (sign-initialised means), ``components`` independent (s, r, b) sets used
round-robin -- the ensemble members of Dusenberry et al.'s rank-1 BNNs.  Same constructor arguments and
attributes (``layer``, ``s``, ``r``, ``bias``, ``component_counter``) as the reference's ``Rank1Linear``
/ ``Rank1Conv2D``; the draws run the HIP draw kernel (forward + analytic backward), the GEMM / conv is
stock PyTorch.  ``BBBOptimizer(components=...)`` divides the data loss accordingly (bbb.py:80).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .util import GaussianParameter


class _Rank1Base(nn.Module):
    """``layer`` is any bias-free nn.Module with a ``weight``; ``_expand`` reshapes a per-channel vector so that
    it broadcasts over the layer's input / output."""

    def _setup(self, layer: nn.Module, n_in: int, n_out: int, bias: bool, components: int, gp_kwargs: dict):
        self.layer = layer
        self.components = components
        self.s = nn.ModuleList([GaussianParameter(n_in, **gp_kwargs) for _ in range(components)])
        self.r = nn.ModuleList([GaussianParameter(n_out, **gp_kwargs) for _ in range(components)])
        self.bias = nn.Parameter(torch.empty((components, n_out))) if bias else None
        self.component_counter = 0
        self.reset_parameters()

    def reset_parameters(self):
        self.layer.reset_parameters()
        for gp in list(self.s) + list(self.r):
            gp.sign_init()
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.layer.weight)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def _expand(self, v: torch.Tensor) -> torch.Tensor:
        return v

    def forward(self, input):
        c = self.component_counter
        s, r = self._expand(self.s[c].sample()), self._expand(self.r[c].sample())
        output = self.layer(input * s) * r
        if self.bias is not None:
            output = output + self._expand(self.bias[c])
        self.component_counter = (c + 1) % self.components
        return output


class Rank1Linear(_Rank1Base):
    def __init__(self, in_features, out_features, prior, bias=True, components=1, **gp_kwargs):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self._setup(nn.Linear(in_features, out_features, bias=False), in_features, out_features, bias, components,
                    gp_kwargs)


class Rank1Conv2D(_Rank1Base):
    def __init__(self, in_channels, out_channels, kernel_size, prior, stride=1, padding=0, bias=True, components=1,
                 **gp_kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False)
        self._setup(conv, in_channels, out_channels, bias, components, gp_kwargs)

    def _expand(self, v: torch.Tensor) -> torch.Tensor:
        return v.unsqueeze(-1).unsqueeze(-1)          # [C] -> [C, 1, 1]: broadcast over H x W


def make_module_rank1(module: nn.Module, prior, components: int, **gp_kwargs) -> int:
    """Swap every nn.Linear / nn.Conv2d (square kernels) of ``module`` for its rank-1 counterpart, keeping the
    trained weight matrix as the deterministic part and the bias as every component's bias (cf. the reference's
    ``make_module_rank1``).  Returns the number of layers replaced."""
    count = 0
    for name, child in list(module.named_children()):
        if isinstance(child, nn.Conv2d):
            new = Rank1Conv2D(child.in_channels, child.out_channels, child.kernel_size[0], prior, stride=child.stride,
                              padding=child.padding, bias=child.bias is not None, components=components, **gp_kwargs)
        elif isinstance(child, nn.Linear):
            new = Rank1Linear(child.in_features, child.out_features, prior, bias=child.bias is not None,
                              components=components, **gp_kwargs)
        else:
            count += make_module_rank1(child, prior, components, **gp_kwargs)
            continue
        new = new.to(child.weight.device)
        with torch.no_grad():
            new.layer.weight.copy_(child.weight)
            if child.bias is not None:
                new.bias.copy_(child.bias.detach().unsqueeze(0).expand_as(new.bias))
        setattr(module, name, new)
        count += 1
    return count
