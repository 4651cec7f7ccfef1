"""DeepEnsemble ("MultiX"): models together with their optimizers.

Reference: ``src/algos/ensemble.py:8-48``.  Same constructor, ``state_dict``
layout and ``predict`` sample split (``samples // members`` each, member 0
takes the remainder).  New: ``predict(..., rank=, world_size=)`` fans the
(member, sample) units out over GPUs (MultiSWAG: 5 modes x 30 samples over 8
MI355X) -- every unit keeps its position in the reference's output order, so
concatenating the ranks' outputs in unit order reproduces the single-process
result.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn as nn


def split_samples(samples: int, members: int) -> List[int]:
    """ensemble.py:37-40: ``samples // members`` each, member 0 gets the rest."""
    per = samples // members
    return [samples - (members - 1) * per] + [per] * (members - 1)


def fan_out(samples: int, members: int, rank: int, world_size: int) -> List[Tuple[int, int, int]]:
    """The (unit index, member, sample-within-member) triples this rank owns:
    units are numbered in the reference's output order and dealt round-robin."""
    units = []
    u = 0
    for member, count in enumerate(split_samples(samples, members)):
        for s in range(count):
            if u % world_size == rank:
                units.append((u, member, s))
            u += 1
    return units


class DeepEnsemble(nn.Module):
    '''
        Stores modules together with their optimizers (drop-in for src/algos/ensemble.py:8-48)
    '''

    def __init__(self, models_and_optimizers):
        super().__init__()
        self.models = nn.ModuleList(list(map(lambda p: p[0], models_and_optimizers)))
        self.optimizers = list(map(lambda p: p[1], models_and_optimizers))

    def state_dict(self, prefix='', keep_vars=False):
        return {
            "models": self.models.state_dict(prefix=prefix, keep_vars=keep_vars),
            "optimizers": list(map(lambda o: o.state_dict(), self.optimizers)),
        }

    def load_state_dict(self, state_dict, strict=True):
        self.models.load_state_dict(state_dict["models"], strict=strict)
        for optimizer, optimizer_state in zip(self.optimizers, state_dict["optimizers"]):
            optimizer.load_state_dict(optimizer_state)

    def predict(self, predict_closure, samples, multisample=False, *, rank=0, world_size=1):
        '''
            Makes <samples> predictions with this ensemble.  predict_closure takes a model of this ensemble
            and makes a single prediction with it; sample_parameters() is called here (ensemble.py:28-44).

            With world_size > 1 only this rank's share of the (member, sample) units is evaluated; the
            result holds them in unit order (see fan_out()).
        '''
        if len(self.models) == 1 and getattr(self.models[0], "supports_multisample", False) and multisample:
            return predict_closure(self.models[0], n_samples=samples)

        output = []
        if world_size == 1:
            for (model, optimizer), model_samples in zip(self.models_and_optimizers,
                                                         split_samples(samples, len(self.models))):
                prefetch = getattr(optimizer, "prefetch_samples", None)
                if prefetch is not None:
                    prefetch(model_samples)      # SWAG, rng="philox": all of this member's samples in one pass
                for _ in range(model_samples):
                    optimizer.sample_parameters()
                    output.append(predict_closure(model))
        else:
            pairs = self.models_and_optimizers
            for _, member, s in fan_out(samples, len(self.models), rank, world_size):
                model, optimizer = pairs[member]
                # per-(member, sample) RNG stream so the result is independent of the GPU count
                if hasattr(optimizer, "_sample_counter"):
                    optimizer._sample_counter = s
                optimizer.sample_parameters()
                output.append(predict_closure(model))
        return torch.stack(output)

    def predict_distributed(self, predict_closure, samples, process_group=None):
        """MultiX fan-out over the ranks of ``process_group`` (one GPU per rank): every rank evaluates its
        share of the (member, sample) units, then ONE all-gather of the (small) prediction tensors puts the
        full ``[samples, ...]`` result, in the reference's output order, on every rank.  Callers reduce it
        exactly as they reduce ``predict()``'s output (e.g. ``logsumexp(out, 0) - log(S)``, camelyon.py:29-30)."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        local = self.predict(predict_closure, samples, rank=rank, world_size=world) if fan_out(
            samples, len(self.models), rank, world) else None
        counts = [len(fan_out(samples, len(self.models), r, world)) for r in range(world)]
        most = max(counts)
        # every rank needs the per-unit shape to size its padded contribution
        shape_src = counts.index(most)
        if local is None:
            unit_shape = None
        else:
            unit_shape = list(local.shape[1:])
        shapes = [None] * world
        dist.all_gather_object(shapes, (unit_shape, str(local.dtype) if local is not None else None), group=process_group)
        unit_shape, dtype_name = shapes[shape_src]
        dtype = getattr(torch, dtype_name.split(".")[-1])
        device = local.device if local is not None else next(self.models.parameters()).device
        padded = torch.zeros([most] + unit_shape, dtype=dtype, device=device)
        if local is not None:
            padded[:local.shape[0]] = local
        gathered = torch.empty([world * most] + unit_shape, dtype=dtype, device=device)
        dist.all_gather_into_tensor(gathered, padded, group=process_group)
        total = sum(counts)
        out = torch.empty([total] + unit_shape, dtype=dtype, device=device)
        for r in range(world):
            for j, (u, _, _) in enumerate(fan_out(samples, len(self.models), r, world)):
                out[u] = gathered[r * most + j]
        return out

    @property
    def models_and_optimizers(self):
        return list(zip(self.models, self.optimizers))
