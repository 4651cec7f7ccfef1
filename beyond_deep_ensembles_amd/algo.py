"""BayesianOptimizer base: the drop-in boundary.

Mirrors the API of the reference's ``src/algos/algo.py:5-80``
(``BayesianOptimizer``) and ``:83-133`` (``LastLayerBayesianOptimizer``):
``step(forward_closure, backward_closure[, grad_scaler])``,
``complete_epoch()``, ``sample_parameters()``, ``init_grad_scaler()``,
``get_base_optimizer()``.  The subclasses keep their statistics in FLAT device
buffers and make the model's parameters views into them (the reference itself
re-points ``param.data`` at optimizer-owned storage: ``svgd.py:127``,
``swag.py:58``, ``ivorn.py:111``), so the HIP kernels see one coalesced tensor
while the user's closures stay untouched.
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional, Sequence

import torch
from torch.optim import Optimizer

from .ops import HipOps, pad4

ROW_HEADER = 16   # spare floats kept behind the D parameters of every flat row


def _default_ops():
    """The kernel backend.  There is exactly one: libbde_hip.so.  Creating it
    raises if the library has not been built."""
    return HipOps()


class FlatLayout:
    """Offsets of a list of parameters inside one flat fp32 row."""

    def __init__(self, params: Sequence[torch.Tensor]):
        self.shapes = [tuple(p.shape) for p in params]
        self.numels = [p.numel() for p in params]
        self.offsets = []
        off = 0
        for n in self.numels:
            self.offsets.append(off)
            off += n
        self.d = off
        # >= ROW_HEADER spare floats per row, rows 256-byte aligned
        self.ld = pad4(self.d + ROW_HEADER)

    def views(self, row: torch.Tensor) -> List[torch.Tensor]:
        """Per-parameter views into a flat row (no copies)."""
        return [row[o:o + n].view(s) for o, n, s in zip(self.offsets, self.numels, self.shapes)]


def check_params(params: Sequence[torch.Tensor], ops) -> None:
    for p in params:
        if p.dtype != torch.float32:
            raise TypeError(f"beyond_deep_ensembles_amd optimizers need float32 parameters, got {p.dtype}")
    if getattr(ops, "name", "") == "hip":
        for p in params:
            if not p.is_cuda:
                raise RuntimeError("beyond_deep_ensembles_amd optimizers need CUDA (HIP) parameters: the posterior "
                                   "updates are HIP kernels with no CPU path (got a parameter on %s)" % p.device)


def adopt_grads(params: Sequence[torch.nn.Parameter], views: Sequence[torch.Tensor], add: bool = False) -> None:
    """After a backward pass make sure the gradients sit in ``views`` (rows of
    the flat gradient buffer).  Autograd accumulates in place into an existing
    ``.grad`` (we point it at the view beforehand), so normally this is only a
    pointer comparison per tensor; if the closure replaced ``.grad`` the values
    are copied over (``add=False``) or added on top of what the view already
    holds (``add=True``) with one multi-tensor op."""
    src, dst = [], []
    for p, v in zip(params, views):
        g = p.grad
        if g is v:            # in-place accumulation keeps the very tensor object we installed (the common case)
            continue
        if g is None:
            if not add:
                v.zero_()
            p.grad = v
        elif g.data_ptr() != v.data_ptr():
            src.append(g)
            dst.append(v)
            p.grad = v
    if src:
        if add:
            torch._foreach_add_(dst, src)
        else:
            torch._foreach_copy_(dst, src)


class BayesianOptimizer(Optimizer):
    '''
        An optimizer that optimizes a distribution over the parameters of a model (approximate inference).
        Same contract as the reference's BayesianOptimizer (src/algos/algo.py:5-17):

        Use the optimizer returned by get_base_optimizer() for learning rate schedulers.
        If you are using a GradScaler, call optimizer.init_grad_scaler(grad_scaler) before the first step.
    '''

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._step_supports_amp_scaling = True

    def step(self, forward_closure, backward_closure):
        '''
            Makes a single step (algo.py:19-29).  forward_closure evaluates the loss for the current state of
            the module and must neither clear gradients nor call backward(); backward_closure runs one backward
            pass (call scale() on the loss there if you use a GradScaler).
        '''
        raise NotImplementedError()

    def complete_epoch(self):
        '''Completes a training epoch (algo.py:31-35).'''
        pass

    def sample_parameters(self):
        '''Samples concrete values for all parameters; call before forward() during evaluation (algo.py:37-42).'''
        raise NotImplementedError()

    def init_grad_scaler(self, grad_scaler):
        '''GradScalers initialise lazily, but their scale is needed before the first step (algo.py:44-49).'''
        if grad_scaler is not None and grad_scaler.is_enabled() and grad_scaler._scale is None:
            grad_scaler._lazy_init_scale_growth_tracker(self._params_device())

    def get_base_optimizer(self):
        '''The optimizer that does the actual parameter updates (algo.py:51-55).'''
        pass

    def _params_device(self):
        return self.param_groups[0]["params"][0].device

    def _params(self):
        for group in self.param_groups:
            for param in group["params"]:
                yield param

    def _prepare_and_check_grads(self, grad_scaler, optimizer=None):
        # algo.py:65-73, including its quirk: the inf check reads
        # self.state["found_inf_per_device"], a fresh empty entry, so it never fails.
        if grad_scaler is None or not grad_scaler.is_enabled():
            return True
        opt = self if optimizer is None else optimizer
        grad_scaler.unscale_(opt)
        return sum(v.item() for v in self.state["found_inf_per_device"].values()) == 0

    def _set_grad_scaler_state(self, grad_scaler, stage, optimizer=None):
        if grad_scaler is None or not grad_scaler.is_enabled():
            return
        opt = self if optimizer is None else optimizer
        grad_scaler._per_optimizer_states[id(opt)]["stage"] = stage


def _opt_state():
    from torch.amp.grad_scaler import OptState
    return OptState


class LastLayerBayesianOptimizer(BayesianOptimizer):
    '''
        Joins a Bayesian optimizer for the last layer(s) with a deterministic optimizer for the rest of the
        network (algo.py:83-133).  Behaviour is undefined if the two optimizers share parameters.
    '''

    def __init__(self, ll_bayesian_optimizer: BayesianOptimizer, deterministic_optimizer: Optimizer):
        self.ll_bayesian_optimizer = ll_bayesian_optimizer
        self.deterministic_optimizer = deterministic_optimizer

    def step(self, forward_closure, backward_closure, grad_scaler=None):
        if grad_scaler is not None and grad_scaler.is_enabled():
            raise ValueError("Doesn't support grad scaler")
        self.deterministic_optimizer.zero_grad()
        # makes at least one forward & backward pass, which creates the gradients of the deterministic part
        loss = self.ll_bayesian_optimizer.step(forward_closure, backward_closure)
        self.deterministic_optimizer.step()
        return loss

    def complete_epoch(self):
        self.ll_bayesian_optimizer.complete_epoch()

    def sample_parameters(self):
        self.ll_bayesian_optimizer.sample_parameters()

    def init_grad_scaler(self, grad_scaler):
        if grad_scaler.is_enabled():
            raise RuntimeError("Doesn't support grad scaler")

    def get_base_optimizer(self):
        raise RuntimeError("There is no defined base optimizer on the ll optimizer. Call get_base_optimizer "
                           "directly on the passed ll bayesian optimizer")

    def state_dict(self) -> Dict[str, Any]:
        return {
            "ll_bayesian_optimizer": self.ll_bayesian_optimizer.state_dict(),
            "deterministic_optimizer": self.deterministic_optimizer.state_dict(),
        }

    def load_state_dict(self, state_dict: Dict[str, Any]) -> None:
        self.ll_bayesian_optimizer.load_state_dict(state_dict["ll_bayesian_optimizer"])
        self.deterministic_optimizer.load_state_dict(state_dict["deterministic_optimizer"])

    def __repr__(self) -> str:
        return ("LL Bayesian Optimizer: \n\n" + self.ll_bayesian_optimizer.__repr__()
                + "\n==================================\nDeterministic Optimizer:\n\n"
                + self.deterministic_optimizer.__repr__())
