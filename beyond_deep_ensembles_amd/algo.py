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
    """Offsets of a list of parameters inside one flat fp32 row.  ``align`` (in floats) pads every tensor's offset up to
    a multiple of it: with ``align=4`` each tensor starts on a 16-byte boundary of the row, so a kernel can read the
    tensor's elements EITHER from the row OR from any other 16-byte-aligned buffer (the gradient tensor autograd
    produced) with the same float4 accesses.  The padding columns belong to no parameter and stay zero."""

    def __init__(self, params: Sequence[torch.Tensor], align: int = 1):
        self.shapes = [tuple(p.shape) for p in params]
        self.numels = [p.numel() for p in params]
        self.offsets = []
        off = 0
        for n in self.numels:
            off = (off + align - 1) // align * align
            self.offsets.append(off)
            off += n
        off = (off + align - 1) // align * align
        self.d = off                                   # columns in use, padding included
        self.n_valid = sum(self.numels)                # columns that belong to a parameter
        # padding BETWEEN tensors (then the parameters' elements are not one contiguous range of the row); padding
        # only behind the last tensor leaves columns [0, n_valid) exactly the parameters' elements
        self.padded = any(o != sum(self.numels[:k]) for k, o in enumerate(self.offsets)) if len(self.numels) < 4096 \
            else self.d != self.n_valid
        self._valid_index = None
        # >= ROW_HEADER spare floats per row, rows 256-byte aligned
        self.ld = pad4(self.d + ROW_HEADER)
        from . import _host
        native = _host.load()
        self._native = native.Layout(self.offsets, self.numels, [list(s) for s in self.shapes]) \
            if native is not None and hasattr(native, "Layout") else None

    def valid_index(self, device) -> torch.Tensor:
        """Column indices of the parameters' elements, in parameter order (the padding skipped)."""
        if self._valid_index is None or self._valid_index.device != torch.device(device):
            idx = torch.cat([torch.arange(o, o + n) for o, n in zip(self.offsets, self.numels)]) if self.numels \
                else torch.zeros(0, dtype=torch.long)
            self._valid_index = idx.to(device)
        return self._valid_index

    def compact(self, rows: torch.Tensor) -> torch.Tensor:
        """``rows [..., >= d]`` -> the parameters' elements only ``[..., n_valid]`` (a view when nothing is padded)."""
        if not self.padded:
            return rows[..., :self.n_valid]
        return rows.index_select(-1, self.valid_index(rows.device))

    def views(self, row: torch.Tensor) -> List[torch.Tensor]:
        """Per-parameter views into a flat row (no copies)."""
        if self._native is not None:
            return self._native.views(row)
        return [row[o:o + n].view(s) for o, n, s in zip(self.offsets, self.numels, self.shapes)]

    def point_data(self, params: Sequence[torch.nn.Parameter], row: torch.Tensor) -> None:
        """``param.data = its view of row`` for every parameter: ``vector_to_parameters`` without the copy
        (``swag.py:58``), one call for the whole list."""
        if self._native is not None:
            self._native.point_data(params, row)
            return
        for p, v in zip(params, self.views(row)):
            p.data = v


def check_params(params: Sequence[torch.Tensor], ops) -> None:
    for p in params:
        if p.dtype != torch.float32:
            raise TypeError(f"beyond_deep_ensembles_amd optimizers need float32 parameters, got {p.dtype}")
    if getattr(ops, "name", "") == "hip":
        for p in params:
            if not p.is_cuda:
                raise RuntimeError("beyond_deep_ensembles_amd optimizers need CUDA (HIP) parameters: the posterior "
                                   "updates are HIP kernels with no CPU path (got a parameter on %s)" % p.device)


def repoint(params: Sequence[torch.nn.Parameter], datas: Optional[Sequence[torch.Tensor]],
            grads: Optional[Sequence[torch.Tensor]] = None) -> None:
    """``param.data = datas[i]`` and/or ``param.grad = grads[i]`` for every parameter: the pointer aliasing the
    reference does per tensor in Python (``svgd.py:127``, ``swag.py:58,81``, ``ivorn.py:111``).  An optimizer step
    re-points 2 * particle_count * n_tensors tensors, which dominates the host time of a step at ResNet-50's 161
    tensors, so the loop runs in the native helper (csrc/host.cpp) when that is built."""
    from . import _host
    native = _host.load()
    if native is not None:
        native.repoint(params, datas, grads)
        return
    if datas is not None:
        for p, v in zip(params, datas):
            p.data = v
    if grads is not None:
        for p, g in zip(params, grads):
            p.grad = g


def clear_grads(params: Sequence[torch.nn.Parameter]) -> None:
    """``param.grad = None`` for every parameter (``zero_grad(set_to_none=True)``).  With no gradient installed
    autograd simply keeps the tensor a backward pass produces; with one installed it launches an in-place add PER
    TENSOR (161 extra launches per backward at ResNet-50's tensor count), which is why the shells clear the
    gradients before every backward pass and move the results into the flat buffers with ONE multi-tensor op."""
    from . import _host
    native = _host.load()
    if native is not None:
        native.clear_grads(params)
        return
    for p in params:
        p.grad = None


def adopt_grads(params: Sequence[torch.nn.Parameter], views: Sequence[torch.Tensor], add: bool = False) -> None:
    """After a backward pass move the gradients into ``views`` (rows of the flat gradient buffer) and make them
    the parameters' ``.grad``: copied (``add=False``; a missing gradient zeroes its view) or added on top of what
    the view already holds (``add=True``), with one multi-tensor op for the whole list.  Gradients that already
    live in their view (in-place accumulation) are left alone."""
    from . import _host
    native = _host.load()
    if native is not None:
        native.adopt_grads(params, views, bool(add))
        return
    src, dst, missing = [], [], []
    for p, v in zip(params, views):
        g = p.grad
        if g is v:
            continue
        if g is None:
            if not add:
                missing.append(v)
        elif g.data_ptr() != v.data_ptr():
            src.append(g)
            dst.append(v)
    if missing:
        torch._foreach_zero_(missing)
    if src:
        if add:
            torch._foreach_add_(dst, src)
        else:
            torch._foreach_copy_(dst, src)
    repoint(params, None, views)


def collect_grads(params: Sequence[torch.nn.Parameter], views: Sequence[torch.Tensor], table: torch.Tensor, j: int,
                  m: int, zero_addr: int = 0) -> List[torch.Tensor]:
    """``_store_grads`` (``svgd.py:129-133``) without the clones: record WHERE each parameter's gradient lives.
    ``table[i * m + j]`` (host int64; i = parameter, j = particle) receives the address of the gradient tensor autograd
    produced when the update kernels can read it in place (fp32, contiguous, 16-byte aligned, on the views' device);
    otherwise the gradient is copied into -- a missing one zeroes -- its view of the flat gradient row and the view's
    address is recorded.  With ``zero_addr`` (the address of a read-only all-zero buffer at least as long as the largest
    tensor) a MISSING gradient costs nothing: that address is recorded and nothing is written.  Returns the tensors
    recorded by reference (keep them alive until the update is enqueued)."""
    from . import _host
    native = _host.load()
    if native is not None and hasattr(native, "collect_grads"):
        return native.collect_grads(params, views, table, int(j), int(m), int(zero_addr))
    keep, src, dst, missing = [], [], [], []
    addrs = []
    for p, v in zip(params, views):
        g = p.grad
        addr = v.data_ptr()
        if g is None:
            if zero_addr:
                addr = zero_addr
            else:
                missing.append(v)
        elif g.data_ptr() == v.data_ptr():
            pass
        elif g.dtype == torch.float32 and g.layout == torch.strided and g.is_contiguous() and g.device == v.device \
                and g.numel() == v.numel() and g.data_ptr() % 16 == 0:
            addr = g.data_ptr()
            keep.append(g)
        else:
            src.append(g)
            dst.append(v)
        addrs.append(addr)
    table.view(-1, m)[:len(addrs), j] = torch.tensor(addrs, dtype=torch.int64)
    with torch.no_grad():
        if missing:
            torch._foreach_zero_(missing)
        if src:
            torch._foreach_copy_(dst, src)
    return keep


def _light_step(fn):
    """torch wraps every Optimizer subclass's ``step`` (Optimizer.profile_hook_step): a profiler range, the step pre /
    post hooks, ``_optimizer_step_code``.  With no hook registered and no profiler running that wrapper is pure host
    cost -- 30-60 us per call on the GPU boxes, a third of a small model's whole SVGD step -- so ``step`` takes it only
    when it has something to do: a hook on this optimizer, a global optimizer hook, or an active autograd profiler.
    (``hooked = True`` is the marker torch itself checks before wrapping.)"""
    import functools
    import torch.optim.optimizer as _o
    full = Optimizer.profile_hook_step(fn)

    @functools.wraps(fn)
    def step(self, *args, **kwargs):
        hooks = getattr(self, "_optimizer_step_pre_hooks", None) is not None      # LastLayerBayesianOptimizer has no Optimizer state
        if hooks and (self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks or _o._global_optimizer_pre_hooks or
                      _o._global_optimizer_post_hooks or torch.autograd._profiler_enabled()):
            return full(self, *args, **kwargs)
        return fn(self, *args, **kwargs)
    step.hooked = True
    return step


class BayesianOptimizer(Optimizer):
    """Base of the posterior-inference optimizers; the public surface is the reference's
    (``src/algos/algo.py:5-80``):

    ==========================================  ==================================================
    ``step(forward_closure, backward_closure)``  one update; the forward closure returns the loss and
                                                 must not clear gradients or call backward, the
                                                 backward closure runs ``loss.backward()`` (or
                                                 ``scaler.scale(loss).backward()``)
    ``complete_epoch()``                         end-of-epoch bookkeeping
    ``sample_parameters()``                      draw concrete weights before an eval forward
    ``get_base_optimizer()``                     what LR schedulers must be attached to
    ``init_grad_scaler(scaler)``                 call once before the first step when using AMP
    ==========================================  ==================================================
    """

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        if "step" in cls.__dict__ and not getattr(cls.__dict__["step"], "hooked", False):
            cls.step = _light_step(cls.__dict__["step"])

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._step_supports_amp_scaling = True     # lets torch.amp.GradScaler.step() pass its kwargs through

    def _keep_live_base_optimizer(self, live) -> None:
        """After ``load_state_dict``: the reference keeps the whole base optimizer OBJECT in ``self.state`` (``svgd.py:51``,
        ``swag.py:28``, ``bbb.py:53``), so a loaded state carries the checkpoint's copy of it -- bound to the checkpoint's
        copies of the parameters, not to this model's (stepping it would train tensors nobody looks at; in the reference a
        resumed run does exactly that).  Here the optimizer this shell was CONSTRUCTED with stays in charge and takes over
        the loaded one's state (momentum / Adam moments, step counts) and hyper-parameters, matched by parameter position as
        ``torch.optim.Optimizer.state_dict`` does.  Checkpoints written for evaluation are unaffected."""
        loaded = self.state.get("__base_optimizer")
        if live is None or loaded is None or loaded is live or not hasattr(loaded, "state_dict"):
            return
        try:
            import copy
            live.load_state_dict(copy.deepcopy(loaded.state_dict()))     # (a state handed over in-process must not stay shared)
        except (ValueError, KeyError, RuntimeError) as e:  # another optimizer class / parameter structure: keep what was loaded
            import warnings
            warnings.warn(
                f"{type(self).__name__}.load_state_dict: the checkpoint's base optimizer ({type(loaded).__name__}) does not fit "
                f"the one this optimizer was constructed with ({type(live).__name__}): {type(e).__name__}: {e}.  The loaded "
                "object stays in charge; it is bound to the checkpoint's copies of the parameters, so continuing to TRAIN "
                "with it will not move this model (evaluation is unaffected).  Construct the optimizer with a base "
                "optimizer of the checkpoint's class and parameter structure to resume training.", RuntimeWarning,
                stacklevel=3)
            return
        self.state["__base_optimizer"] = live

    # -- the reference's abstract surface -----------------------------------
    def step(self, forward_closure, backward_closure):
        raise NotImplementedError()

    def sample_parameters(self):
        raise NotImplementedError()

    def complete_epoch(self):
        pass

    def get_base_optimizer(self):
        pass

    def init_grad_scaler(self, grad_scaler):
        # a GradScaler creates its scale tensor lazily, but step() needs it before the first unscale
        needs_init = grad_scaler is not None and grad_scaler.is_enabled() and grad_scaler._scale is None
        if needs_init:
            grad_scaler._lazy_init_scale_growth_tracker(self._params_device())

    # -- helpers shared by the subclasses ------------------------------------
    def _params(self):
        for group in self.param_groups:
            yield from group["params"]

    def _params_device(self):
        return self.param_groups[0]["params"][0].device

    @staticmethod
    def _scaler_active(grad_scaler) -> bool:
        return grad_scaler is not None and grad_scaler.is_enabled()

    def _prepare_and_check_grads(self, grad_scaler, optimizer=None):
        """Unscale the gradients of ``optimizer`` (default: self) and report whether they are finite.
        Mirrors algo.py:65-73 INCLUDING its quirk: the check reads
        ``self.state["found_inf_per_device"]``, an empty defaultdict entry, so it always passes."""
        if not self._scaler_active(grad_scaler):
            return True
        grad_scaler.unscale_(optimizer if optimizer is not None else self)
        found = self.state["found_inf_per_device"]
        return sum(v.item() for v in found.values()) == 0

    def _set_grad_scaler_state(self, grad_scaler, stage, optimizer=None):
        """Force the GradScaler's per-optimizer state machine (READY / UNSCALED / STEPPED), algo.py:75-80."""
        if self._scaler_active(grad_scaler):
            target = optimizer if optimizer is not None else self
            grad_scaler._per_optimizer_states[id(target)]["stage"] = stage


def _opt_state():
    from torch.amp.grad_scaler import OptState
    return OptState


class LastLayerBayesianOptimizer(BayesianOptimizer):
    """A Bayesian optimizer for the head and a plain torch optimizer for the rest of the network, stepped
    together (``src/algos/algo.py:83-133``).  The two must not share parameters.  GradScalers are not
    supported, exactly as in the reference."""

    def __init__(self, ll_bayesian_optimizer: BayesianOptimizer, deterministic_optimizer: Optimizer):
        # deliberately no Optimizer.__init__: this object owns no parameters of its own
        self.ll_bayesian_optimizer = ll_bayesian_optimizer
        self.deterministic_optimizer = deterministic_optimizer

    @property
    def _parts(self):
        return {"ll_bayesian_optimizer": self.ll_bayesian_optimizer,
                "deterministic_optimizer": self.deterministic_optimizer}

    def step(self, forward_closure, backward_closure, grad_scaler=None):
        if BayesianOptimizer._scaler_active(grad_scaler):
            raise ValueError("Doesn't support grad scaler")
        self.deterministic_optimizer.zero_grad()
        # the Bayesian part runs >= 1 forward/backward pass, which also fills the backbone's gradients
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
        return {name: part.state_dict() for name, part in self._parts.items()}

    def load_state_dict(self, state_dict: Dict[str, Any]) -> None:
        for name, part in self._parts.items():
            part.load_state_dict(state_dict[name])

    def __repr__(self) -> str:
        bar = "=" * 34
        return f"LL Bayesian Optimizer: \n\n{self.ll_bayesian_optimizer!r}\n{bar}\nDeterministic Optimizer:\n\n{self.deterministic_optimizer!r}"
