"""Stein Variational Gradient Descent behind the reference's SVGDOptimizer API.

Reference: ``src/algos/svgd.py`` -- ``rbf`` (:14-32) and ``SVGDOptimizer``
(:37-135).  Same constructor, ``step`` / ``sample_parameters`` /
``get_base_optimizer`` behaviour and ``self.state`` keys; what changes is where
the data lives and who does the arithmetic:

* the M particles are rows of ONE flat device buffer ``P [M, ld]`` and their
  gradients rows of ``G [M, ld]``; ``state[param]["particle_i"]`` are views
  into ``P`` (the reference keeps M * n_tensors separate clones and re-gathers
  them with stack/cat every step, svgd.py:83-84);
* during the M forward/backward passes ``param.grad`` is a view of row i of
  ``G``, so autograd accumulates straight into the flat buffer (no
  ``_store_grads`` clones, svgd.py:129-133);
* the posterior update (svgd.py:86-89) is three HIP launches
  (``bde_svgd_step``: MFMA Gram, bandwidth/kernel statistics, streaming
  combine) that leave ``-phi`` in ``G``, whose rows then ARE the gradients the
  shared base optimizer consumes (svgd.py:92-103; no per-tensor clones).

Multi-GPU (not in the reference): with ``process_group`` each rank runs the
forward/backward passes of its own M/W particles; the gradient rows are
exchanged with ONE RCCL all-gather (xGMI) into the replicated ``G``, then
every rank applies the same deterministic update to its replica of ``P``.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from .algo import BayesianOptimizer, FlatLayout, adopt_grads, check_params, _default_ops, _opt_state


def rbf(particles: torch.Tensor, h_override=None, _ops=None):
    """Pairwise RBF kernel with the median heuristic and its repulsive gradient
    (drop-in for ``src/algos/svgd.py:14-32``): returns ``(kernel [M, M],
    grad_kernel [M, D])`` for ``particles [M, D]`` on the GPU."""
    ops = _ops or _default_ops()
    m, d = particles.shape
    layout_ld = (d + 63) // 64 * 64
    if particles.stride(0) == layout_ld and particles.stride(1) == 1 and particles.data_ptr() % 16 == 0:
        P = particles
    else:
        P = particles.new_zeros((m, layout_ld))
        P[:, :d] = particles
    ws, ks = ops.svgd_ws(m, P.device), ops.svgd_kstat(m, P.device)
    out = torch.zeros_like(P)
    ops.svgd_gram(P, d, ws)
    ops.svgd_kstats(ws, m, 0.0, 1.0, 1.0, 1.0, ks, h_override=float(h_override) if h_override is not None else 0.0,
                    mode=1)
    ops.svgd_combine(P, None, out, d, ks)
    return ks[:m * m].view(m, m).clone(), out[:, :d]


class SVGDOptimizer(BayesianOptimizer):
    '''
        Stein Variational Gradient Descent (drop-in for src/algos/svgd.py:37-135).

        This optimizer does not support multiple parameter groups, as they are used to differentiate between
        the particles.  The base optimizer must optimize the model's parameters; its state is therefore shared
        by all particles and advanced particle_count times per step, exactly as in the reference.

        Extra keyword-only arguments (not in the reference):
          process_group       shard the particles' forward/backward passes over the ranks of this group
          fuse_base_optimizer apply a torch.optim.SGD / Adam base optimizer inside the update kernel (one pass over
                              P and G that writes the updated particles; -phi is never materialised)
          reuse_gram          with fuse_base_optimizer: the fused kernel also emits the Gram partials of the updated
                              particles, so the next step skips the Gram pass.  Only valid while nothing but this
                              optimizer modifies the particles between two steps (call invalidate_gram() otherwise).
    '''

    def __init__(self, params, reset_params_closure, base_optimizer, particle_count, dataset_size, l2_reg=0.0,
                 kernel_grad_scale=1.0, *, process_group=None, fuse_base_optimizer=False, reuse_gram=False, _ops=None):
        super().__init__(map(lambda p: {"params": p}, params), {})
        self._ops = _ops or _default_ops()
        self.state["__base_optimizer"] = base_optimizer
        self.state["__l2_reg"] = l2_reg
        self.state["__dataset_size"] = dataset_size
        self.state["__current_particle"] = 0
        self.state["__particle_count"] = particle_count
        self.state["__kernel_grad_scale"] = kernel_grad_scale

        plist = list(self._params())
        check_params(plist, self._ops)
        self._plist = plist
        self._layout = FlatLayout(plist)
        dev = self._params_device()
        m, ld = particle_count, self._layout.ld
        # flat particle / gradient storage; padding stays zero
        self._P = torch.zeros((m, ld), dtype=torch.float32, device=dev)
        self._G = torch.zeros((m, ld), dtype=torch.float32, device=dev)
        self._pviews: List[List[torch.Tensor]] = [self._layout.views(self._P[i]) for i in range(m)]
        self._gviews: List[List[torch.Tensor]] = [self._layout.views(self._G[i]) for i in range(m)]
        self._ws = self._ops.svgd_ws(m, dev)
        self._kstat = self._ops.svgd_kstat(m, dev)

        # particle 0 = the current weights, particles 1.. = after reset_params_closure() (svgd.py:54-59)
        for particle_idx in range(particle_count):
            with torch.no_grad():
                torch._foreach_copy_(self._pviews[particle_idx], [p.detach() for p in plist])
            for param, view in zip(plist, self._pviews[particle_idx]):
                self.state[param][f"particle_{particle_idx}"] = view
            if particle_idx < particle_count - 1:
                reset_params_closure()

        # ---- multi-GPU sharding of the particles (new) ----
        self._pg = process_group
        self._world, self._rank = 1, 0
        if process_group is not None:
            import torch.distributed as dist
            self._world, self._rank = dist.get_world_size(process_group), dist.get_rank(process_group)
            if particle_count % self._world != 0:
                raise ValueError(f"particle_count ({particle_count}) must be a multiple of the group size ({self._world})")
            # identical particles on every rank whatever the local RNG state was
            dist.broadcast(self._P, src=dist.get_global_rank(process_group, 0), group=process_group)
        self._fuse = bool(fuse_base_optimizer) and particle_count <= 16     # fused kernels: single-tile path only
        self._tmp = None
        self._fused_state = None
        self._reuse_gram = bool(reuse_gram) and self._fuse and self._ops.svgd_fused_gram_supported(particle_count)
        self._gram_valid = False

    # ------------------------------------------------------------------
    def _local_particles(self) -> range:
        per = self.state["__particle_count"] // self._world
        return range(self._rank * per, (self._rank + 1) * per)

    def step(self, forward_closure, backward_closure, grad_scaler=None):
        OptState = _opt_state()
        base = self.state["__base_optimizer"]
        m, d = self.state["__particle_count"], self._layout.d
        total_loss = torch.tensor(0.0, device=self._params_device())
        for particle_idx in self._local_particles():
            self._set_grad_scaler_state(grad_scaler, OptState.READY, base)
            # _use_particle (svgd.py:120-127) and base_optimizer.zero_grad() (svgd.py:70) in one pass over the
            # tensors: the gradient row is zeroed and param.grad pointed at it, so backward() accumulates
            # into the flat buffer
            self._G[particle_idx].zero_()
            for param, pview, gview in zip(self._plist, self._pviews[particle_idx], self._gviews[particle_idx]):
                param.data = pview
                param.grad = gview

            loss = forward_closure()
            total_loss += loss.detach()
            backward_closure(loss)
            if not self._prepare_and_check_grads(grad_scaler, base):
                return None
            adopt_grads(self._plist, self._gviews[particle_idx])      # _store_grads (svgd.py:129-133)

        with torch.no_grad():
            pending = self._start_gradient_exchange(total_loss) if self._world > 1 else None
            fused = self._fuse and (grad_scaler is None or not grad_scaler.is_enabled())
            # The Gram pass and the kernel statistics need only the (replicated) particles: they run while the
            # gradient all-gather is in flight.  (Skipped when the previous fused kernel already left the Gram.)
            if not (fused and self._reuse_gram and self._gram_valid):
                self._ops.svgd_gram(self._P, d, self._ws)
            self._ops.svgd_kstats(self._ws, m, float(self.state["__l2_reg"]), float(self.state["__kernel_grad_scale"]),
                                  float(self.state["__dataset_size"]), -1.0, self._kstat)
            if pending is not None:
                total_loss = self._finish_gradient_exchange(pending)
            if fused:
                # ONE pass: -phi in registers, M shared-state optimizer applications, updated particles out
                self._fused_apply(base)
                self._gram_valid = self._reuse_gram
                self._use_particle(m - 1)    # the reference leaves the model aliased to the last particle
            else:
                self._gram_valid = False
                # svgd.py:86-89: -phi overwrites the gradient rows
                if m <= 16:
                    self._ops.svgd_combine(self._P, self._G, self._G, d, self._kstat)
                else:
                    # the blocked path for > 16 particles produces 16 rows per pass and re-reads all of G
                    if self._tmp is None:
                        self._tmp = torch.zeros_like(self._G)
                    self._ops.svgd_combine(self._P, self._G, self._tmp, d, self._kstat)
                    self._G.copy_(self._tmp)
                # write the modified gradients TO THE ORIGINAL PARAMETERS and call the optimizer on them (svgd.py:92-103)
                for particle_idx in range(m):
                    for model_param, pview, gview in zip(self._plist, self._pviews[particle_idx], self._gviews[particle_idx]):
                        model_param.grad = gview
                        model_param.data = pview
                    if grad_scaler is not None:
                        self._set_grad_scaler_state(grad_scaler, OptState.UNSCALED, base)
                        grad_scaler.step(base)
                    else:
                        base.step()

        return total_loss / self.state["__particle_count"]

    # ------------------------------------------------------------------
    def _start_gradient_exchange(self, local_loss_sum: torch.Tensor):
        """ONE all-gather of the gradient rows (RCCL over xGMI on the GPU box), asynchronous.
        The particle's loss rides in the spare floats behind the D gradients of its row, so no
        second collective is needed for the returned loss."""
        import torch.distributed as dist
        m, d = self.state["__particle_count"], self._layout.d
        per = m // self._world
        lo = self._rank * per
        # every local particle's row carries (sum of local losses / per) -> summing all rows / M = mean loss
        self._G[lo:lo + per, d] = local_loss_sum / per
        own = self._G[lo:lo + per].reshape(-1)
        if dist.get_backend(self._pg) == "gloo":
            own = own.clone()                  # gloo wants disjoint input/output
        return dist.all_gather_into_tensor(self._G.view(-1), own, group=self._pg, async_op=True)

    def _finish_gradient_exchange(self, work) -> torch.Tensor:
        d = self._layout.d
        work.wait()
        total = self._G[:, d].sum()
        self._G[:, d] = 0
        return total

    def _fused_apply(self, base) -> None:
        """-phi and the M sequential base-optimizer applications with shared state in ONE
        kernel (bde_svgd_fused_sgd / bde_svgd_fused_adam); hyper-parameters are read from the
        base optimizer's param_groups every step, so LR schedulers keep working."""
        groups = base.param_groups
        g0 = groups[0]
        keys = [k for k in g0 if k != "params"]
        for g in groups[1:]:
            if any(g[k] != g0[k] for k in keys):
                raise RuntimeError("fuse_base_optimizer needs identical hyper-parameters in all param groups")
        d, ld = self._layout.d, self._layout.ld
        dev = self._P.device
        if isinstance(base, torch.optim.SGD):
            if g0.get("maximize", False):
                raise RuntimeError("fuse_base_optimizer: maximize=True is not supported")
            if self._fused_state is None:
                self._fused_state = self.state["__fused"] = {"kind": "sgd", "buf": torch.zeros(ld, device=dev), "first": True}
            st = self._fused_state
            self._ops.svgd_fused_sgd(self._P, self._G, st["buf"], d, self._kstat, g0["lr"], g0["momentum"],
                                     g0["dampening"], g0["weight_decay"], g0["nesterov"], st["first"],
                                     ws_next=self._ws if self._reuse_gram else None)
            st["first"] = False
        elif type(base) is torch.optim.Adam:
            if g0.get("amsgrad", False) or g0.get("maximize", False):
                raise RuntimeError("fuse_base_optimizer: amsgrad / maximize are not supported")
            if self._fused_state is None:
                self._fused_state = self.state["__fused"] = {"kind": "adam", "exp_avg": torch.zeros(ld, device=dev),
                                                             "exp_avg_sq": torch.zeros(ld, device=dev), "step": 0}
            st = self._fused_state
            lr = g0["lr"]
            self._ops.svgd_fused_adam(self._P, self._G, st["exp_avg"], st["exp_avg_sq"], d, self._kstat, float(lr),
                                      g0["betas"][0], g0["betas"][1], g0["eps"], g0["weight_decay"], st["step"],
                                      ws_next=self._ws if self._reuse_gram else None)
            st["step"] += self.state["__particle_count"]
        else:
            raise RuntimeError(f"fuse_base_optimizer supports torch.optim.SGD and torch.optim.Adam, got {type(base)}")

    # ------------------------------------------------------------------
    def sample_parameters(self):
        '''Cycles through the particles (svgd.py:107-112)'''
        self._use_particle(self.state["__current_particle"])
        self.state["__current_particle"] = (self.state["__current_particle"] + 1) % self.state["__particle_count"]

    def _params_for_particle(self, particle_idx):
        particle = f"particle_{particle_idx}"
        for group in self.param_groups:
            for param in group["params"]:
                yield self.state[param][particle]

    def _use_particle(self, particle_idx):
        '''Does *not* clone: updates of the model parameters are updates of the particle (svgd.py:120-127)'''
        for param, view in zip(self._plist, self._pviews[particle_idx]):
            param.data = view

    def get_base_optimizer(self):
        return self.state["__base_optimizer"]

    # ---- flat access (bench / multi-GPU tests / checkpoints) -----------
    @property
    def particles(self) -> torch.Tensor:
        """[M, D] view of the flat particle buffer."""
        return self._P[:, :self._layout.d]

    @property
    def kernel_stats(self) -> dict:
        m = self.state["__particle_count"]
        ks = self._kstat
        return {"kernel": ks[:m * m].view(m, m), "d2": ks[m * m:2 * m * m].view(m, m),
                "h": ks[2 * m * m + m], "median": ks[2 * m * m + m + 1]}

    def invalidate_gram(self) -> None:
        """Call after modifying the particles outside this optimizer when reuse_gram=True."""
        self._gram_valid = False

    def load_state_dict(self, state_dict):
        """Accepts the reference's layout (per-tensor ``particle_i`` entries): the
        values are copied into the flat buffer and the state re-aliased to it."""
        super().load_state_dict(state_dict)
        self._gram_valid = False
        self._fused_state = self.state.get("__fused")          # shared optimizer state of the fused path
        if self._fused_state is not None:
            for k, v in self._fused_state.items():
                if torch.is_tensor(v):
                    self._fused_state[k] = v.to(self._P.device)
        m = self.state["__particle_count"]
        with torch.no_grad():
            for i in range(m):
                for param, view in zip(self._plist, self._pviews[i]):
                    loaded = self.state[param][f"particle_{i}"]
                    if loaded.data_ptr() != view.data_ptr():
                        view.copy_(loaded)
                    self.state[param][f"particle_{i}"] = view
