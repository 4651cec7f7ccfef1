"""Improved Variational Online Newton behind the reference's iVONOptimizer API.

Reference: ``src/algos/ivorn.py:7-127``.  Same constructor, step /
sample_parameters / get_base_optimizer behaviour and per-parameter state keys
(``mean``, ``momentum``, ``precision``, ``delta``, ``acc_grad`` -- here views
into flat per-group buffers).  What changes: the weight-noise draw
(ivorn.py:102-115, ~6 ATen launches per tensor per MC sample) is one kernel,
the update block (ivorn.py:76-89, ~14 launches per tensor) is one fused kernel
(32 B/param), and the MC gradients are summed into the flat gradient buffer with
one multi-tensor op per MC sample (``param.grad`` is cleared before every
backward pass so autograd hands over fresh tensors), which replaces
``_store_gradients`` (ivorn.py:120-127).

Noise: ``rng="torch"`` (default) draws one ``normal_like`` per tensor in
parameter order, i.e. consumes the reference's random stream;
``rng="philox"`` generates the noise inside the kernel.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from .algo import BayesianOptimizer, FlatLayout, adopt_grads, check_params, clear_grads, repoint, _default_ops, _opt_state
from .util import normal_like


class _Group:
    def __init__(self, group, device):
        self.params = list(group["params"])
        self.layout = FlatLayout(self.params)
        ld = self.layout.ld

        def buf(fill=0.0):
            return torch.full((ld,), fill, dtype=torch.float32, device=device)

        self.theta, self.mean, self.momentum = buf(), buf(), buf()
        self.precision = buf(group["prior_prec"] / group["N"])      # ivorn.py:34 (N without augmentation)
        self.delta, self.grad, self.eps = buf(), buf(), None
        self.theta_views = self.layout.views(self.theta)
        self.grad_views = self.layout.views(self.grad)
        with torch.no_grad():
            torch._foreach_copy_(self.theta_views, [p.detach() for p in self.params])
            self.mean.copy_(self.theta)
        self.have_delta = False


class iVONOptimizer(BayesianOptimizer):
    '''
        Improved Variational Online Newton (drop-in for src/algos/ivorn.py:7-127)
    '''

    def __init__(self, params, lr, prior_prec, dataset_size, betas=(0.9, 0.999), damping=0.0, tempering=1.0,
                 augmentation=1.0, mc_samples=5, deterministic=False, *, rng="torch", seed=0, _ops=None):
        defaults = {
            "lr": lr,
            "betas": betas,
            "prior_prec": prior_prec,
            "damping": damping,
            "tempering": tempering,
            "augmentation": augmentation,
            "N": dataset_size,
            "deterministic": deterministic,
            "step": 0,
        }
        super().__init__(params, defaults)
        self._ops = _ops or _default_ops()
        if rng not in ("torch", "philox"):
            raise ValueError("rng must be 'torch' or 'philox'")
        self.rng = rng
        self.seed = int(seed)
        self.noise_source: Optional[Callable[[int], torch.Tensor]] = None   # parity tests: flat eps of one group
        self._draw_counter = 0
        check_params(list(self._params()), self._ops)
        dev = self._params_device()
        self._groups: List[_Group] = [_Group(g, dev) for g in self.param_groups]
        for fg in self._groups:
            mviews, moviews, pviews = (fg.layout.views(b) for b in (fg.mean, fg.momentum, fg.precision))
            dviews = fg.layout.views(fg.delta)
            for i, param in enumerate(fg.params):
                param.data = fg.theta_views[i]
                state = self.state[param]
                state["mean"], state["momentum"], state["precision"] = mviews[i], moviews[i], pviews[i]
                state["delta"], state["acc_grad"] = dviews[i], fg.grad_views[i]

        assert mc_samples > 0
        self.mc_samples = mc_samples

    def step(self, forward_closure, backward_closure, grad_scaler=None):
        OptState = _opt_state()
        self._reset_state()
        scaler_on = grad_scaler is not None and grad_scaler.is_enabled()

        acc_loss = None
        for mc in range(self.mc_samples):
            # READY so that the GradScaler does not complain when calling unscale_ (ivorn.py:46-47)
            self._set_grad_scaler_state(grad_scaler, OptState.READY)

            self.sample_parameters()
            with torch.enable_grad():
                # no gradient installed: backward() hands over fresh tensors (no per-tensor in-place add launches);
                # _store_gradients moves / adds them into the flat accumulator with one multi-tensor op
                for fg in self._groups:
                    clear_grads(fg.params)
                loss = forward_closure()
                backward_closure(loss)

            if acc_loss is None:
                acc_loss = loss
            else:
                acc_loss += loss

            if not self._prepare_and_check_grads(grad_scaler):
                return None

            self._store_gradients(scaler_on, first=(mc == 0))
        acc_loss /= self.mc_samples

        with torch.no_grad():
            for group, fg in zip(self.param_groups, self._groups):
                group["step"] += 1
                t = group["step"]
                beta1, beta2 = group["betas"]
                n_eff = group["N"] * group["augmentation"]                       # ivorn.py:72
                lam = group["tempering"] * group["prior_prec"] / n_eff           # ivorn.py:74
                self._ops.ivon_update(fg.mean, fg.momentum, fg.precision, fg.delta, fg.grad, fg.layout.d, lam=lam,
                                      n_eff=n_eff, mc=self.mc_samples, beta1=beta1, beta2=beta2, t=t, lr=group["lr"],
                                      damping=group["damping"])

        self._set_grad_scaler_state(grad_scaler, OptState.STEPPED)
        return acc_loss

    def _reset_state(self):
        for fg in self._groups:
            fg.have_delta = False

    def sample_parameters(self):
        with torch.no_grad():
            for group, fg in zip(self.param_groups, self._groups):
                n_eff = group["N"] * group["augmentation"]
                d = fg.layout.d
                eps = None
                if not group["deterministic"]:
                    if self.noise_source is not None:
                        eps = self.noise_source(d)
                        if eps.numel() != fg.layout.ld:
                            padded = torch.zeros(fg.layout.ld, dtype=torch.float32, device=fg.theta.device)
                            padded[:d] = eps
                            eps = padded
                    elif self.rng == "torch":
                        # one normal_like per tensor, in parameter order (ivorn.py:108)
                        if fg.eps is None:
                            fg.eps = torch.zeros_like(fg.theta)
                        for v in fg.layout.views(fg.eps):
                            v.normal_(0, 1)
                        eps = fg.eps
                self._ops.ivon_sample(fg.mean, fg.precision, fg.theta, fg.delta, d, n_eff, first=not fg.have_delta,
                                      eps=eps, seed=self.seed, stream_id=self._draw_counter,
                                      deterministic=bool(group["deterministic"]))
                fg.have_delta = True
                self._draw_counter += 1
                repoint(fg.params, fg.theta_views, None)

    def get_base_optimizer(self):
        return self

    def load_state_dict(self, state_dict):
        """Accepts the reference's layout (per-tensor ``mean`` / ``momentum`` / ``precision`` entries):
        the values are copied into the flat buffers and the state re-aliased to their views."""
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for fg in self._groups:
                views = {"mean": fg.layout.views(fg.mean), "momentum": fg.layout.views(fg.momentum),
                         "precision": fg.layout.views(fg.precision), "delta": fg.layout.views(fg.delta),
                         "acc_grad": fg.grad_views}
                for i, param in enumerate(fg.params):
                    state = self.state[param]
                    for key, vs in views.items():
                        loaded = state.get(key)
                        if torch.is_tensor(loaded) and loaded.data_ptr() != vs[i].data_ptr():
                            vs[i].copy_(loaded)
                        state[key] = vs[i]
                    # the model's weights were loaded by the model's own state_dict: adopt them as theta
                    fg.theta_views[i].copy_(param.detach())
                    param.data = fg.theta_views[i]
                fg.have_delta = False

    def _store_gradients(self, scaler_on=False, first=True):
        """ivorn.py:120-127: the first MC sample's gradients become the accumulator, the others are added."""
        for fg in self._groups:
            adopt_grads(fg.params, fg.grad_views, add=not first)
