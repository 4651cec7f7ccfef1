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
    """The (unit index, member, sample-within-member) triples this rank owns.  Units are numbered in the
    reference's output order (member 0's samples, then member 1's, ...) and every rank takes ONE contiguous
    range of them, so a rank touches at most two neighbouring members when there are at least as many units per
    member as per rank (MultiSWAG 5 x 30 over 8 GPUs: 18-19 units per rank): it needs only those members'
    statistics in HBM, and each of its member blocks is one pass of the batched sampler."""
    lo = samples * rank // world_size
    hi = samples * (rank + 1) // world_size
    units = []
    u = 0
    for member, count in enumerate(split_samples(samples, members)):
        for s in range(count):
            if lo <= u < hi:
                units.append((u, member, s))
            u += 1
    return units


def members_needed(samples: int, members: int, rank: int, world_size: int) -> List[int]:
    """Indices of the ensemble members whose posterior this rank samples from (see ``fan_out``)."""
    return sorted({member for _, member, _ in fan_out(samples, members, rank, world_size)})


class DeepEnsemble(nn.Module):
    '''
        Stores modules together with their optimizers (drop-in for src/algos/ensemble.py:8-48)
    '''

    def __init__(self, models_and_optimizers):
        super().__init__()
        pairs = list(models_and_optimizers)
        # the models are registered sub-modules (their weights appear in parameters() / state_dict()); the
        # optimizers are plain attributes, kept in the same order (ensemble.py:12-15)
        self.models = nn.ModuleList([model for model, _ in pairs])
        self.optimizers = [optimizer for _, optimizer in pairs]

    @property
    def models_and_optimizers(self):
        return list(zip(self.models, self.optimizers))

    # checkpoint layout of the reference (ensemble.py:17-26): {"models": ModuleList state, "optimizers": [state, ...]}
    def state_dict(self, prefix='', keep_vars=False):
        optimizer_states = [optimizer.state_dict() for optimizer in self.optimizers]
        return {"models": self.models.state_dict(prefix=prefix, keep_vars=keep_vars), "optimizers": optimizer_states}

    def load_state_dict(self, state_dict, strict=True):
        self.models.load_state_dict(state_dict["models"], strict=strict)
        for optimizer, saved in zip(self.optimizers, state_dict["optimizers"]):
            optimizer.load_state_dict(saved)

    def predict(self, predict_closure, samples, multisample=False, *, rank=0, world_size=1):
        '''
            Makes <samples> predictions with this ensemble.  predict_closure takes a model of this ensemble
            and makes a single prediction with it; sample_parameters() is called here (ensemble.py:28-44).

            With world_size > 1 only this rank's share of the (member, sample) units is evaluated; the
            result holds them in unit order (see fan_out()).  Every member's sampler ends where the
            single-process call would leave it, on every rank, whatever the world size.
        '''
        if len(self.models) == 1 and getattr(self.models[0], "supports_multisample", False) and multisample:
            return predict_closure(self.models[0], n_samples=samples)

        counts = split_samples(samples, len(self.models))
        mine = fan_out(samples, len(self.models), rank, world_size)
        output = []
        for member, ((model, optimizer), count) in enumerate(zip(self.models_and_optimizers, counts)):
            own = [s for _, mem, s in mine if mem == member]           # a contiguous range of this member's samples
            start = _sampler_position(optimizer)
            if own:
                _seek_sampler(optimizer, start, own[0])
                prefetch = getattr(optimizer, "prefetch_samples", None)
                if prefetch is not None:
                    prefetch(len(own))           # SWAG, rng="philox": this block's samples in one batched MFMA pass
                for _ in own:
                    optimizer.sample_parameters()
                    output.append(predict_closure(model))
            _seek_sampler(optimizer, start, count)                     # where the single-process loop ends
        return torch.stack(output)

    def predict_distributed(self, predict_closure, samples, process_group=None):
        """MultiX fan-out over the ranks of ``process_group`` (one GPU per rank): every rank evaluates its
        share of the (member, sample) units, then ONE all-gather of the (small) prediction tensors puts the
        full ``[samples, ...]`` result, in the reference's output order, on every rank.  Callers reduce it
        exactly as they reduce ``predict()``'s output (e.g. ``logsumexp(out, 0) - log(S)``, camelyon.py:29-30).

        Every rank owns at least one unit whenever ``samples >= world size`` and then knows the shape of a
        prediction from its own output; only a call with fewer samples than ranks needs one small object
        all-gather beside it."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        n_members = len(self.models)
        plan = self._fan_out_plan(samples, n_members, world)
        counts, most = plan["counts"], plan["most"]
        local = self.predict(predict_closure, samples, rank=rank, world_size=world) if counts[rank] else None
        unit_shape, dtype = (list(local.shape[1:]), local.dtype) if local is not None else (None, None)
        if min(counts) == 0:
            # some rank has nothing to evaluate and therefore does not know what a prediction looks like
            shapes = [None] * world
            dist.all_gather_object(shapes, (unit_shape, str(dtype) if dtype is not None else None), group=process_group)
            unit_shape, dtype_name = shapes[counts.index(most)]
            dtype = getattr(torch, dtype_name.split(".")[-1])
        device = local.device if local is not None else next(self.models.parameters()).device
        if local is not None and local.shape[0] == most:
            padded = local.contiguous()
        else:
            padded = torch.zeros([most] + unit_shape, dtype=dtype, device=device)
            if local is not None:
                padded[:local.shape[0]] = local
        gathered = torch.empty([world * most] + unit_shape, dtype=dtype, device=device)
        dist.all_gather_into_tensor(gathered, padded, group=process_group)
        if plan["dense"]:
            return gathered                          # every rank holds `most` units: already in unit order
        if plan["take"].device != device:
            plan["take"] = plan["take"].to(device)
        return gathered.index_select(0, plan["take"])   # drop the padding rows: one gather, unit order

    def _fan_out_plan(self, samples, n_members, world):
        """Rows of the all-gathered (padded) prediction tensor in unit order, per (samples, members, world): computed once."""
        key = (samples, n_members, world)
        cache = self.__dict__.setdefault("_plans", {})
        if key not in cache:
            counts = [len(fan_out(samples, n_members, r, world)) for r in range(world)]
            most = max(counts)
            # ranks own contiguous unit ranges in rank order, so unit order = rank order with the padding rows skipped
            take = [r * most + j for r in range(world) for j in range(counts[r])]
            cache[key] = {"counts": counts, "most": most, "dense": all(c == most for c in counts),
                          "take": torch.tensor(take, dtype=torch.long)}
        return cache[key]


def _sampler_position(optimizer):
    """Where the optimizer's sample sequence stands: the Philox stream counter of a SWAG optimizer, the
    round-robin particle index of an SVGD optimizer (svgd.py:107-112); None for samplers without a position."""
    if hasattr(optimizer, "_sample_counter"):
        return optimizer._sample_counter
    state = getattr(optimizer, "state", None)
    if state is not None and "__current_particle" in state:
        return state["__current_particle"]
    return None


def _seek_sampler(optimizer, start, offset):
    """Position the sampler ``offset`` draws after ``start``."""
    if start is None:
        return
    if hasattr(optimizer, "_sample_counter"):
        optimizer._sample_counter = start + offset
        if getattr(optimizer, "_prefetched", None) is not None:
            optimizer._prefetched = None
    else:
        optimizer.state["__current_particle"] = (start + offset) % optimizer.state["__particle_count"]
