"""Stochastic Weight Averaging-Gaussian behind the reference's SwagOptimizer API.

Reference: ``src/algos/swag.py:10-114``.  Same constructor, step /
sample_parameters / complete_epoch behaviour, integer schedule and
``self.state`` keys.  What changes:

* the model's parameters are views into ONE flat device vector ``theta``, so
  the moment update reads the weights in place -- the reference flattens them
  with ``parameters_to_vector(...).cpu()`` every update (swag.py:100);
* ``__mean`` / ``__sq_weights`` stay on the device and ``__deviations`` is a
  ring of K rows (``__dev_head`` = row the next update overwrites) instead of
  a CPU ``[D, K]`` matrix that is physically rolled (swag.py:103).  The K + 2
  statistics rows are the rows of ONE ``[K + 2, ld]`` buffer (ring rows, mean,
  second moment).  ``mean_vector()`` / ``deviations_dk()`` / ``state_dict()``
  give the reference's layout back;
* a posterior sample is one kernel (``bde_swag_sample``) that writes straight
  into a second flat vector the parameters are re-pointed at; restoring the
  training weights (swag.py:76-82) is a pointer swap, not a clone;
* no ``LowRankMultivariateNormal`` object is built: its constructor's
  capacitance matrix / Cholesky is never needed to SAMPLE (swag.py:114).

Noise: ``rng="torch"`` (default) draws ``eps_W [K]`` then ``eps_D [D]`` from
torch's generator exactly as ``rsample`` would on this device, so a seeded run
consumes the reference's random stream; ``rng="philox"`` generates the noise
inside the kernel (no extra HBM traffic; stream = sample counter).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Tuple

import torch

from .algo import BayesianOptimizer, FlatLayout, check_params, repoint, _default_ops


class SwagOptimizer(BayesianOptimizer):
    '''
        Stochastic Weight Averaging-Gaussian (drop-in for src/algos/swag.py:10-114)
    '''

    def __init__(self, params, base_optimizer, update_interval, start_epoch=0, deviation_samples=30, *,
                 rng="torch", seed=0, _ops=None):
        super().__init__(params, {})
        self._ops = _ops or _default_ops()
        if rng not in ("torch", "philox"):
            raise ValueError("rng must be 'torch' or 'philox'")
        if not 1 <= int(deviation_samples) <= 256:
            raise ValueError(f"deviation_samples must be in [1, 256] (BDE_MAX_RANK of the sampling kernels), got "
                             f"{deviation_samples}")

        self.start_epoch = start_epoch
        self.update_interval = math.floor(update_interval)
        self.param_dist = None                      # kept for attribute compatibility; never built
        self.deviation_samples = deviation_samples
        self.rng = rng
        self.seed = int(seed)
        self.noise_source: Optional[Callable[[int, int], Tuple[torch.Tensor, torch.Tensor]]] = None
        self._sample_counter = 0
        self._prefetched = None          # (rows [n, ld], next row): samples generated ahead by prefetch_samples()
        self._points_at = "theta"        # which buffer the parameters currently view: "theta" | "sample" | "row"

        plist = list(self._params())
        check_params(plist, self._ops)
        self._plist = plist
        self._layout = FlatLayout(plist)
        dev = self._params_device()
        d, ld, k = self._layout.d, self._layout.ld, deviation_samples
        # serving a small model's prefetched sample: re-point n_tensors views (~1 us of host time each) or copy the row
        self._copy_is_cheaper = len(plist) * 1e-6 > 8.0 * d / 5e12

        # flat training weights; the parameters become views of it
        self._theta = torch.zeros(ld, dtype=torch.float32, device=dev)
        self._theta_views = self._layout.views(self._theta)
        with torch.no_grad():
            torch._foreach_copy_(self._theta_views, [p.detach() for p in plist])
        # flat target of posterior samples
        self._sample = torch.zeros(ld, dtype=torch.float32, device=dev)
        self._sample_views = self._layout.views(self._sample)
        for param, view in zip(plist, self._theta_views):
            param.data = view
            self.state[param]["original_param"] = view          # swag.py:26 (a clone there)

        self.state["__base_optimizer"] = base_optimizer
        self._live_base = base_optimizer
        self.state["__epoch"] = 0
        self.state["__steps_since_swag_start"] = 0
        self.state["__updates"] = 0
        # rows 0 .. K-1: the deviation ring (swag.py:34), row K: mean (swag.py:32: the initial weights are sample #1),
        # row K + 1: second moment (swag.py:33)
        self._stats = torch.zeros((k + 2, ld), dtype=torch.float32, device=dev)
        with torch.no_grad():
            self._stats[k, :d] = self._theta[:d]
            self._stats[k + 1, :d] = self._theta[:d] ** 2
        self._publish_stats()
        self.state["__dev_head"] = 0
        self.state["__params_dirty"] = False

    def _publish_stats(self) -> None:
        """The reference's state keys, as views of the statistics buffer (``__deviations``: the ring rows ``[K, ld]``)."""
        k = self.deviation_samples
        self.state["__mean"] = self._stats[k]
        self.state["__sq_weights"] = self._stats[k + 1]
        self.state["__deviations"] = self._stats[:k]

    def _stat_rows(self):
        """(mean, sq, ring) for the kernels."""
        k = self.deviation_samples
        return self._stats[k], self._stats[k + 1], self._stats[:k]

    # ------------------------------------------------------------------
    def step(self, forward_closure, backward_closure, grad_scaler=None):
        """One ordinary training step of the wrapped optimizer, then the moment-collection gate (swag.py:37-51).
        Training always continues from the training weights: if the model still carries a posterior sample from an
        evaluation, the parameters are pointed back first."""
        base = self.state["__base_optimizer"]
        self._prefetched = None                     # samples drawn ahead belong to the posterior before this step
        self._restore_original_params()
        base.zero_grad()
        loss = forward_closure()
        backward_closure(loss)
        if grad_scaler is None:
            base.step()
        else:
            grad_scaler.step(base)                  # unscales, skips the step on inf/nan gradients
        self._swag_update()
        return loss

    def prefetch_samples(self, n_samples: int, max_bytes: int = 8 << 30) -> int:
        """Generate the next ``n_samples`` posterior samples in ONE pass over the statistics
        (bde_swag_sample_batched, MFMA low-rank product, 4*D*(K+2+S) bytes instead of S*4*D*(K+3));
        the following ``sample_parameters()`` calls then only re-point the parameters at the stored rows.
        Only with ``rng="philox"`` (stream id = sample counter, so the samples are the ones the unbatched
        kernel would produce); otherwise a no-op.  Returns the number of samples prefetched.
        ``DeepEnsemble.predict`` calls this for every member."""
        if self.rng != "philox" or self.noise_source is not None or n_samples < 2:
            return 0
        n = int(min(n_samples, max(1, max_bytes // (4 * self._layout.ld))))
        d = self._layout.d
        mean, sq, ring = self._stat_rows()
        rows = torch.empty((n, self._layout.ld), dtype=torch.float32, device=self._params_device())
        with torch.no_grad():
            for lo in range(0, n, 32):
                hi = min(n, lo + 32)
                self._ops.swag_sample_batched(mean, sq, ring, self.state["__dev_head"], rows[lo:hi], d,
                                              seed=self.seed, stream_id0=self._sample_counter + lo)
        self._prefetched = [rows, 0]
        return n

    def sample_parameters(self):
        self._save_original_params()
        self.state["__params_dirty"] = True
        if self._prefetched is not None:
            rows, nxt = self._prefetched
            n_rows = rows.shape[0]
            if self._copy_is_cheaper:
                # many tensors: one device copy of the row into the sample vector the parameters already view
                # (8 D bytes of HBM traffic) beats re-pointing every tensor (~1 us of host time each)
                with torch.no_grad():
                    self._sample.copy_(rows[nxt])
                self._point_at_sample_vector()
            else:
                self._layout.point_data(self._plist, rows[nxt])
                self._points_at = "row"
            self._sample_counter += 1
            self._prefetched = [rows, nxt + 1] if nxt + 1 < n_rows else None
            return
        d, k = self._layout.d, self.deviation_samples
        eps_w = eps_d = None
        if self.noise_source is not None:
            eps_w, eps_d = self.noise_source(k, d)
            eps_d = self._pad(eps_d)
        elif self.rng == "torch":
            # rsample() draws eps_W first, then eps_D (torch LowRankMultivariateNormal.rsample)
            dev = self._params_device()
            eps_w = torch.empty(k, dtype=torch.float32, device=dev).normal_()
            eps_d = torch.empty(d, dtype=torch.float32, device=dev).normal_()
            eps_d = self._pad(eps_d)
        mean, sq, ring = self._stat_rows()
        with torch.no_grad():
            self._ops.swag_sample(mean, sq, ring, self.state["__dev_head"], self._sample, d, eps_w=eps_w, eps_d=eps_d,
                                  seed=self.seed, stream_id=self._sample_counter)
        self._sample_counter += 1
        # vector_to_parameters (swag.py:58): the parameters become views of the sampled vector
        self._point_at_sample_vector()

    def _point_at_sample_vector(self):
        if self._points_at != "sample":         # consecutive samples land in the same vector: nothing to re-point
            repoint(self._plist, self._sample_views, None)
            self._points_at = "sample"

    def complete_epoch(self):
        self.state["__epoch"] += 1

    def get_base_optimizer(self):
        return self.state["__base_optimizer"]

    # ------------------------------------------------------------------
    def _pad(self, v: torch.Tensor) -> torch.Tensor:
        """16-byte aligned, contiguous noise vector (no copy when it already is)."""
        if v.is_contiguous() and v.data_ptr() % 16 == 0:
            return v
        out = torch.empty(self._layout.ld, dtype=torch.float32, device=v.device)
        out[:v.numel()] = v
        return out

    def _restore_original_params(self):
        if self.state["__params_dirty"]:
            repoint(self._plist, self._theta_views, None)    # swag.py:81 clones; here the weights were never overwritten
            self._points_at = "theta"
            self.state["__params_dirty"] = False

    def _save_original_params(self):
        # swag.py:84-89: the training weights live in self._theta and sampling never writes there
        pass

    def _swag_update(self):
        if self.state["__epoch"] >= self.start_epoch:
            self.state["__steps_since_swag_start"] += 1

            if self.state["__steps_since_swag_start"] % self.update_interval == 0:
                assert not self.state["__params_dirty"]
                with torch.no_grad():
                    self.state["__updates"] += 1
                    updates = self.state["__updates"]
                    head = self.state["__dev_head"]
                    mean, sq, _ = self._stat_rows()
                    self._ops.swag_update(self._theta, mean, sq, self._stats[head], updates, self._layout.d)
                    self.state["__dev_head"] = (head + 1) % self.deviation_samples
                    self.param_dist = None
                    self._prefetched = None

    # ---- reference-layout accessors ------------------------------------
    def mean_vector(self) -> torch.Tensor:
        return self._stats[self.deviation_samples, :self._layout.d]

    def sq_vector(self) -> torch.Tensor:
        return self._stats[self.deviation_samples + 1, :self._layout.d]

    def deviations_dk(self) -> torch.Tensor:
        """The deviation matrix in the reference's layout: ``[D, K]``, oldest
        column first, newest last (what swag.py:103-104 maintains by rolling)."""
        k, head = self.deviation_samples, self.state["__dev_head"]
        order = [(head + c) % k for c in range(k)]
        return self._stats[order, :self._layout.d].t().contiguous()

    def sample_batch(self, n_samples: int, seed: Optional[int] = None, stream_id0: Optional[int] = None) -> torch.Tensor:
        """``n_samples`` posterior samples ``[S, D]`` in one pass over the statistics
        (bde_swag_sample_batched, MFMA low-rank product); Philox noise with streams
        ``stream_id0 + s`` -- identical to ``n_samples`` rng="philox" calls."""
        d, ld = self._layout.d, self._layout.ld
        out = torch.empty((n_samples, ld), dtype=torch.float32, device=self._params_device())
        s0 = self._sample_counter if stream_id0 is None else stream_id0
        mean, sq, ring = self._stat_rows()
        with torch.no_grad():
            for lo in range(0, n_samples, 32):
                hi = min(n_samples, lo + 32)
                self._ops.swag_sample_batched(mean, sq, ring, self.state["__dev_head"], out[lo:hi], d,
                                              seed=self.seed if seed is None else seed, stream_id0=s0 + lo)
        if stream_id0 is None:
            self._sample_counter += n_samples
        return out[:, :d]

    def use_sample(self, flat_sample: torch.Tensor) -> None:
        """Point the parameters at a row of ``sample_batch`` (no copy)."""
        self.state["__params_dirty"] = True
        self._layout.point_data(self._plist, flat_sample)
        self._points_at = "row"

    # ---- checkpoints in the reference's wire layout ---------------------
    def state_dict(self):
        sd = super().state_dict()
        st = sd["state"]
        d = self._layout.d
        st["__mean"] = self.mean_vector().detach().cpu()
        st["__sq_weights"] = self.sq_vector().detach().cpu()
        st["__deviations"] = self.deviations_dk().detach().cpu()
        st.pop("__dev_head", None)
        st["__sample_counter"] = self._sample_counter     # Philox stream position (extra key; the reference ignores it)
        return sd

    def load_state_dict(self, state_dict: dict):
        super().load_state_dict(state_dict)
        self._keep_live_base_optimizer(self._live_base)
        dev, d, ld, k = self._params_device(), self._layout.d, self._layout.ld, self.deviation_samples
        self._prefetched = None                          # rows drawn from the old posterior must not be served
        self._sample_counter = int(self.state.pop("__sample_counter", 0))

        loaded_mean, loaded_sq, devs = self.state["__mean"], self.state["__sq_weights"], self.state["__deviations"]
        self._stats = torch.zeros((k + 2, ld), dtype=torch.float32, device=dev)
        with torch.no_grad():
            self._stats[k, :d] = loaded_mean.reshape(-1)[:d].to(dev).float()
            self._stats[k + 1, :d] = loaded_sq.reshape(-1)[:d].to(dev).float()
            if devs.dim() == 2 and devs.shape[0] == d and devs.shape[1] == k and not (d == k and "__dev_head" in self.state):
                self._stats[:k, :d] = devs.to(dev).float().t()   # reference layout [D, K]: column c -> ring row c, head 0
                self.state["__dev_head"] = 0
            else:
                self._stats[:k, :d] = devs[:, :d].to(dev).float()    # ring rows [K, >= D]
        self._publish_stats()
        # re-alias the parameters to the flat weight vector (values come from the model's own state_dict)
        with torch.no_grad():
            torch._foreach_copy_(self._theta_views, [p.detach() for p in self._plist])
        for param, view in zip(self._plist, self._theta_views):
            param.data = view
            self.state[param]["original_param"] = view
        self.state["__params_dirty"] = False
        self._points_at = "theta"
