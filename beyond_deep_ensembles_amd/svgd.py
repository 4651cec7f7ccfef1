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
* the posterior update (svgd.py:86-89) is three HIP launches (MFMA Gram,
  bandwidth/kernel statistics, streaming combine) -- ONE persistent launch for
  small models -- that leave ``-phi`` in ``G``, whose rows then ARE the
  gradients the shared base optimizer consumes (svgd.py:92-103), or, with
  ``fuse_base_optimizer``, one pass that also applies the optimizer.

Multi-GPU (not in the reference): with ``process_group``
each rank runs the forward/backward passes of its own M/W particles, then

* ``exchange="allgather"`` (default): the gradient rows are exchanged with an
  RCCL all-gather (xGMI) into a replicated ``G`` and every rank applies the same
  deterministic update to its replica of ``P``.  ``exchange_chunks=C`` splits
  the exchange into C column chunks so that the update of chunk c runs while
  chunk c+1 is still on the wire;
* ``exchange="alltoall"``: dimension-sharded.  Every rank owns a column slice
  of ALL particles (and of the shared optimizer state); the gradient rows are
  re-partitioned by an all-to-all, the Gram blocks of the slices are exchanged
  (a few hundred doubles), each rank updates its slice with the fused kernel,
  and a second all-to-all returns the updated slices to the particles' owners:
  2/W of the all-gather's bytes per link and 1/W of the update per rank.
"""
from __future__ import annotations

import os
import warnings
from typing import List, Optional

import numpy as np
import torch

from .algo import (BayesianOptimizer, FlatLayout, adopt_grads, check_params, clear_grads, collect_grads, repoint,
                   _default_ops, _opt_state)
from .ops import pad4


def _raw_stream_of(dev) -> int:
    """hipStream_t of torch's current stream when ``dev`` is the current GPU; 0 for CPU tensors (the tests' backends);
    -1: another GPU than the current one (the caller takes the path that switches devices)."""
    if dev.type != "cuda":
        return 0
    if dev.index is not None and dev.index != torch.cuda.current_device():
        return -1
    from . import ops as _ops_mod
    return int(_ops_mod._stream() or 0)


def rbf(particles: torch.Tensor, h_override=None, _ops=None, _small=None):
    """Pairwise RBF kernel with the median heuristic and its repulsive gradient
    (drop-in for ``src/algos/svgd.py:14-32``): returns ``(kernel [M, M],
    grad_kernel [M, D])`` for ``particles [M, D]`` on the GPU.  ``_small``: True = the small-model kernel where it
    supports the problem, False = the streaming kernels, None = the small-model kernel once it is device-verified
    (device_verified.py)."""
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
    h = float(h_override) if h_override is not None else 0.0
    if _small is None:
        from . import device_verified
        _small = device_verified.enabled("svgd_small")
    if _small and ops.svgd_small_supported(m, d):
        ops.svgd_step_small(P, None, out, d, 0.0, 1.0, 1.0, 1.0, ws, ks, h_override=h, mode=1)   # two small launches
    else:
        ops.svgd_gram(P, d, ws)
        ops.svgd_kstats(ws, m, 0.0, 1.0, 1.0, 1.0, ks, h_override=h, mode=1)
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
          exchange            "allgather" (replicated particles, gradient rows gathered) or "alltoall"
                              (dimension-sharded particles and optimizer state; needs fuse_base_optimizer)
          exchange_chunks     "allgather" only: pipeline the gather and the update over this many column chunks
          overlap_backward    with exchange_chunks > 1: start the gather of a column chunk as soon as the LAST local
                              particle's backward pass has produced the gradients of every tensor in it (chunks leave
                              in a fixed order, last chunk first -- the order backward fills them), so the exchange
                              overlaps the rest of that backward pass like DDP's gradient buckets.  backward_closure
                              must call backward() exactly once and must not touch the gradients afterwards (clipping,
                              scaling, accumulation: a RuntimeError says so); ignored while a GradScaler is active (its
                              unscale pass runs after backward)
          fuse_base_optimizer apply a torch.optim.SGD / Adam base optimizer inside the update kernel (one pass over
                              P and G that writes the updated particles; -phi is never materialised).  "auto" (the
                              DEFAULT: what the reference's constructor call gets): fused
                              exactly when that is indistinguishable from particle_count calls of base.step() -- a
                              plain torch.optim.SGD / Adam (not a subclass) over exactly this optimizer's parameters,
                              one set of hyper-parameters, no amsgrad / maximize / capturable / differentiable /
                              decoupled_weight_decay, no step hooks AT CONSTRUCTION (a hook registered on the base
                              optimizer later makes the next step raise: the fused update never calls base.step(), so it
                              could not run the hook), particle_count <= 64 (and, for 17..64 particles,
                              no chunked or dimension-sharded exchange) -- and the torch loop (False) otherwise; an
                              enabled GradScaler keeps the torch loop as well.  Up to 16 particles:
                              one pass; 17 to 64 (single GPU or exchange="allgather" without chunks): the blocked
                              update kernel writes -phi, then ONE launch applies the base optimizer to all
                              particles in order with its shared state
          reuse_gram          with fuse_base_optimizer: the fused kernel also emits the Gram partials of the updated
                              particles, so the next step skips the Gram pass.  Only valid while nothing but this
                              optimizer modifies the particles between two steps (call invalidate_gram() otherwise).
          single_launch       which kernels update a small model on one GPU (bde_svgd_small_supported: M <= 8,
                              D <= 524,288).  False: the streaming kernels (Gram -> statistics -> combine / fused), as at
                              every other size.  "two": the small-model kernel (bde_svgd_step_small*: the whole update in two
                              launches).  None (default): "two" once that kernel's parity tests have been green on an MI355X
                              for the sources in this tree (device_verified.py, family "svgd_small"), False until then --
                              a default-constructed optimizer only launches device-verified kernels.  (Rounds 2-3 also
                              offered True = one persistent launch with an in-kernel hand-off; it was no faster than the two
                              launches once its wait was bounded, and is gone.)
          host_fast_paths     the native host paths written since the last device run -- the per-particle loop with one
                              native call per particle (ParticleSet.end_begin), the loss mean by one launch
                              (bde_mean_scalars), the small-model step's two C-ABI calls from one native function
                              (host.cpp small_step_*).  None (default): each one only once device_verified.py holds a
                              record for it; True / False: all on / all off (tests, A/B runs).
    '''

    def __init__(self, params, reset_params_closure, base_optimizer, particle_count, dataset_size, l2_reg=0.0,
                 kernel_grad_scale=1.0, *, process_group=None, exchange="allgather", exchange_chunks=1,
                 overlap_backward=False, fuse_base_optimizer="auto", reuse_gram=False, single_launch=None,
                 graph_replay=False, host_fast_paths=None, _ops=None, _force_exchange=False):
        # one param group per tensor, like the reference (svgd.py:50): groups distinguish tensors, not particles
        super().__init__([{"params": p} for p in params], {})
        self._ops = _ops or _default_ops()
        self._live_base = base_optimizer
        for key, value in (("__base_optimizer", base_optimizer), ("__l2_reg", l2_reg), ("__dataset_size", dataset_size),
                           ("__current_particle", 0), ("__particle_count", particle_count),
                           ("__kernel_grad_scale", kernel_grad_scale)):
            self.state[key] = value
        if exchange not in ("allgather", "alltoall"):
            raise ValueError("exchange must be 'allgather' or 'alltoall'")

        plist = list(self._params())
        check_params(plist, self._ops)
        self._plist = plist
        # every tensor starts on a float4 boundary of the row, so the update kernels can read a particle's gradients
        # straight from the tensors autograd produced (no _store_grads copy; see _end_particle)
        self._layout = FlatLayout(plist, align=4)
        dev = self._params_device()
        m, ld = particle_count, self._layout.ld
        # flat particle / gradient storage; padding stays zero
        self._P = torch.zeros((m, ld), dtype=torch.float32, device=dev)
        self._G = torch.zeros((m, ld), dtype=torch.float32, device=dev)
        self._pviews: List[Optional[List[torch.Tensor]]] = [self._layout.views(self._P[i]) for i in range(m)]
        self._gviews: List[Optional[List[torch.Tensor]]] = [self._layout.views(self._G[i]) for i in range(m)]
        self._ws = self._ops.svgd_ws(m, dev)
        self._kstat = self._ops.svgd_kstat(m, dev)
        # where the gradients of the current step live (per tensor and particle), the tensors held by reference, and
        # whether the table describes the step being assembled (step() fills it; a caller that writes the flat
        # gradient rows itself and calls _posterior_update directly does not)
        self._seg = self._ops.seg_table(self._layout.offsets, self._layout.numels, m, dev) \
            if hasattr(self._ops, "seg_table") and m <= 16 else None
        self._seg_host = None
        self._retained = [None] * m
        self._pset = None
        # what a parameter without a gradient points the kernels at (read-only; nothing is zeroed per step)
        self._zeros = torch.zeros(max(self._layout.numels) + 4, dtype=torch.float32, device=dev) if self._seg is not None else None
        if hasattr(self._ops, "load_code_objects"):
            self._ops.load_code_objects(dev)      # every kernel resident on THIS device before any collective / sharing

        # particle 0 = the current weights, every further particle = the weights after one more
        # reset_params_closure() (svgd.py:54-59)
        for particle_idx in range(m):
            if particle_idx > 0:
                reset_params_closure()
            with torch.no_grad():
                torch._foreach_copy_(self._pviews[particle_idx], [p.detach() for p in plist])
            for param, view in zip(plist, self._pviews[particle_idx]):
                self.state[param][f"particle_{particle_idx}"] = view

        # ---- multi-GPU sharding of the particles (new) ----
        self._pg = process_group
        self._world, self._rank = 1, 0
        self._exchange = "allgather"
        self._chunks = None
        # does the update go through the collectives?  A group of ONE rank normally does not (nothing to exchange);
        # ``_force_exchange`` (tests / bench only) sends it through them anyway, so that the RCCL code paths -- the
        # in-place all_gather_into_tensor, all_to_all_single, the chunk pipeline -- execute on a single-GPU box
        self._sharded = False
        if process_group is not None:
            import torch.distributed as dist
            self._world, self._rank = dist.get_world_size(process_group), dist.get_rank(process_group)
            if particle_count % self._world != 0:
                raise ValueError(f"particle_count ({particle_count}) must be a multiple of the group size ({self._world})")
            # identical particles on every rank whatever the local RNG state was
            dist.broadcast(self._P, src=dist.get_global_rank(process_group, 0), group=process_group)
            self._sharded = self._world > 1 or bool(_force_exchange)
            if self._sharded:
                self._exchange = exchange
        if fuse_base_optimizer == "auto":
            fuse_base_optimizer = self._fusable(base_optimizer, plist, particle_count)
        if single_launch is True:
            raise ValueError("single_launch=True (one persistent launch with an in-kernel hand-off) was removed in round 4: it "
                             "was no faster than the two launches of the default; pass None")
        if single_launch not in (None, False, "two"):
            raise ValueError("single_launch must be None, 'two' or False")
        self._single_launch = single_launch
        if host_fast_paths not in (None, True, False):
            raise ValueError("host_fast_paths must be None, True or False")
        self._host_fast_paths = host_fast_paths
        self._gates = {}
        # opt-in: the small-model step's launches (table upload, gradient packing, the two launches of the update) recorded
        # ONCE per set of step scalars in a hipGraph and replayed (see _replay_small_sgd)
        self._graph_replay = bool(graph_replay)
        self._mean_losses = None
        self._small_ok = None
        self._small_step_native = None
        self._graphs, self._graph_eager_steps, self._graph_captures, self._graph_replays = {}, 0, 0, 0
        self._fuse = bool(fuse_base_optimizer) and particle_count <= 16     # fused kernels: single-tile path only
        # 17 <= particle_count <= 64: -phi by the blocked update kernel, then ONE launch applies the base optimizer to all
        # particles in order with its shared state (bde_svgd_apply_sgd / adam) instead of particle_count torch steps
        self._fuse_staged = bool(fuse_base_optimizer) and 16 < particle_count <= 64 and exchange != "alltoall" and \
            int(exchange_chunks) <= 1
        self._fused_decision = None
        self._tmp = None
        self._fused_state = None
        self._reuse_gram = bool(reuse_gram) and self._fuse and self._ops.svgd_fused_gram_supported(particle_count)
        self._gram_valid = False
        if int(exchange_chunks) > 1 and particle_count > 16:
            raise ValueError("exchange_chunks > 1 (pipelined all-gather) needs particle_count <= 16: the blocked update "
                             "for more particles re-reads all gradient rows per pass and cannot consume a staged chunk")
        if self._exchange == "allgather" and self._sharded and int(exchange_chunks) > 1:
            clen = pad4((ld + int(exchange_chunks) - 1) // int(exchange_chunks))
            self._chunks = [(c0, min(ld, c0 + clen)) for c0 in range(0, ld, clen)]
            self._stage = [torch.zeros((m, c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in self._chunks]
        if self._exchange == "alltoall":
            self._init_dimension_sharding()
        self._ov = None                                   # state of an overlapped exchange while a backward pass runs
        self._overlap = bool(overlap_backward) and self._chunks is not None and self._seg is not None
        if self._overlap:
            self._init_overlap()

    # ---- chunked all-gather overlapped with the last local particle's backward pass -------------------------
    def _init_overlap(self):
        """Which tensors a column chunk needs, which table pieces cover it, and one post-accumulate hook per tensor."""
        offs, nums = self._layout.offsets, self._layout.numels
        self._chunk_tensors = [[k for k, (o, n) in enumerate(zip(offs, nums)) if o < c1 and o + n > c0]
                               for c0, c1 in self._chunks]
        self._tensor_chunks = [[] for _ in offs]
        for c, ks in enumerate(self._chunk_tensors):
            for k in ks:
                self._tensor_chunks[k].append(c)
        self._chunk_pieces = [self._seg.piece_range(c0, c1) for c0, c1 in self._chunks]
        for k, p in enumerate(self._plist):
            p.register_post_accumulate_grad_hook(lambda param, k=k: self._grad_ready(k, param))

    def _begin_overlap(self, particle_idx: int, loss_sum: torch.Tensor) -> None:
        """Called between the forward and the backward pass of the last local particle."""
        per = self.state["__particle_count"] // self._world
        if self._seg_host is None:
            self._seg_host = self._seg.staging()
        self._ov = {"row": particle_idx, "pending": [len(ks) for ks in self._chunk_tensors], "seen": {},
                    "next": len(self._chunks) - 1, "works": [None] * len(self._chunks), "loss": loss_sum / per}

    def _grad_ready(self, k: int, param) -> None:
        """Post-accumulate hook of tensor k: record where its gradient lives; chunks whose tensors are all in leave."""
        ov = self._ov
        if ov is None or k in ov["seen"]:
            return
        g, view = param.grad, self._gviews[ov["row"]][k]
        # what left on the wire: the gradient tensor as it is NOW (checked again when backward has returned)
        ov["seen"][k] = (g.data_ptr(), g._version) if g is not None else None
        addr = view.data_ptr()
        if g is not None and g.data_ptr() != addr:
            if g.dtype == torch.float32 and g.layout == torch.strided and g.is_contiguous() and g.device == view.device \
                    and g.numel() == view.numel() and g.data_ptr() % 16 == 0:
                addr = g.data_ptr()
            else:
                with torch.no_grad():
                    view.copy_(g)
        self._seg_host[k * self._seg.m + ov["row"]] = addr
        for c in self._tensor_chunks[k]:
            ov["pending"][c] -= 1
        self._launch_ready_chunks()

    def _launch_ready_chunks(self, force: bool = False) -> None:
        """Chunks leave strictly in descending order (the same order on every rank), each as soon as it is complete."""
        import torch.distributed as dist
        ov = self._ov
        m, d = self.state["__particle_count"], self._layout.d
        per = m // self._world
        lo = self._rank * per
        while ov["next"] >= 0 and (force or ov["pending"][ov["next"]] == 0):
            c = ov["next"]
            ov["next"] -= 1
            (c0, c1), stage = self._chunks[c], self._stage[c]
            with torch.no_grad():
                self._seg.upload_again()
                self._ops.svgd_gather_seg(self._G, self._seg, lo, per, pieces=self._chunk_pieces[c])
                if c0 <= d < c1:
                    self._G[lo:lo + per, d] = ov["loss"]               # the loss rides in the spare floats of the row
                send = self._G[lo, c0:c1] if per == 1 else self._G[lo:lo + per, c0:c1].contiguous().view(-1)
                ov["works"][c] = dist.all_gather_into_tensor(stage.view(-1), send, group=self._pg, async_op=True)

    @staticmethod
    def _fusable(base, plist, particle_count) -> bool:
        """fuse_base_optimizer="auto": True iff the in-kernel SGD / Adam applications are the base optimizer's own."""
        if type(base) not in (torch.optim.SGD, torch.optim.Adam) or particle_count > 64:
            return False
        groups = base.param_groups
        if {id(p) for g in groups for p in g["params"]} != {id(p) for p in plist}:
            return False
        keys = [k for k in groups[0] if k != "params"]
        if any(g[k] != groups[0][k] for g in groups[1:] for k in keys):
            return False
        g0 = groups[0]
        if any(g0.get(flag, False) for flag in ("amsgrad", "maximize", "capturable", "differentiable",
                                                "decoupled_weight_decay")):
            return False
        if any(torch.is_tensor(g0.get(k)) for k in ("lr", "momentum", "weight_decay", "eps")):
            return False
        hooks = ("_optimizer_step_pre_hooks", "_optimizer_step_post_hooks")
        if any(len(getattr(base, h, {}) or {}) for h in hooks):
            return False
        import torch.optim.optimizer as _o                               # hooks registered for ALL optimizers
        return not (getattr(_o, "_global_optimizer_pre_hooks", None) or getattr(_o, "_global_optimizer_post_hooks", None))

    # ------------------------------------------------------------------
    def _local_particles(self) -> range:
        per = self.state["__particle_count"] // self._world
        return range(self._rank * per, (self._rank + 1) * per)

    def _grad_row(self, particle_idx: int) -> torch.Tensor:
        if self._exchange == "alltoall":
            return self._Gown[particle_idx - self._local_particles().start]
        return self._G[particle_idx]

    def step(self, forward_closure, backward_closure, grad_scaler=None):
        local = self._local_particles()
        if grad_scaler is None or not grad_scaler.is_enabled():
            pset = self._particle_set()
            if pset is not None and self._seg is not None and not self._overlap and hasattr(pset, "end_begin") \
                    and self._gate("fast_loop"):
                return self._step_fast(forward_closure, backward_closure, local, pset)
        OptState = _opt_state()
        base = self.state["__base_optimizer"]
        total_loss = None
        last = local[-1]
        for particle_idx in local:
            self._set_grad_scaler_state(grad_scaler, OptState.READY, base)
            self._begin_particle(particle_idx)
            loss = forward_closure()
            # svgd.py:66,72: total_loss = tensor(0.0); total_loss += loss.  0 + x == x bit for bit, so the first loss starts
            # the sum (no host-to-device copy of a constant, one launch less per step)
            if total_loss is None:
                total_loss = loss.detach().to(device=self._params_device(), dtype=torch.float32, copy=True)
            else:
                total_loss += loss.detach()
            if self._overlap and particle_idx == last and not self._scaler_active(grad_scaler):
                self._begin_overlap(particle_idx, total_loss)
            backward_closure(loss)
            if not self._prepare_and_check_grads(grad_scaler, base):
                return None
            self._end_particle(particle_idx)

        return self._posterior_update(total_loss, grad_scaler)

    def _gate(self, family: str) -> bool:
        """May this optimizer take the path of ``family`` (device_verified.FAMILIES)?  host_fast_paths=True / False answers
        for every family; None asks the device-verification table (once per family and optimizer)."""
        hit = self._gates.get(family)
        if hit is None:
            if self._host_fast_paths is not None and family != "svgd_small":
                hit = bool(self._host_fast_paths)
            else:
                from . import device_verified
                hit = device_verified.enabled(family)
            self._gates[family] = hit
        return hit

    def _step_fast(self, forward_closure, backward_closure, local, pset):
        """The same loop for the common case -- no GradScaler, no overlapped exchange, native ParticleSet: nothing but
        the closures, the loss sum and ONE native call per particle (end of particle i + begin of particle i + 1)
        between two forward passes.  A small model's step is host-bound (BENCH extra svgd_step_cifar_resnet20_shell_fused:
        ~115 us per step around 15 us of kernels in round 3; on a stub library in the build container 529 us (round 3) ->
        321 (round 4) -> ~40 (round 5: this loop, the loss mean and the update's launches each ONE native call))."""
        if self._seg_host is None:
            self._seg_host = self._seg.staging()
        table, m_tab, zero = self._seg_host, self._seg.m, self._zeros.data_ptr()
        row_off = local.start if self._exchange == "alltoall" else 0
        first, last = local.start, local.stop - 1
        pset.begin(first)
        losses = []
        for particle_idx in local:
            loss = forward_closure()
            losses.append(loss.detach())
            backward_closure(loss)
            if particle_idx != last:
                pset.end_begin(particle_idx, table, particle_idx - row_off, m_tab, zero, particle_idx + 1)
            else:
                pset.end(particle_idx, table, particle_idx - row_off, m_tab, zero)
        if not self._sharded and self._chunks is None:
            # one GPU: nothing is exchanged, the returned mean (svgd.py:105) comes out of the same launch as the sum
            m = self.state["__particle_count"]
            return self._posterior_update(self._sum_losses(losses, divisor=m), None, mean_taken=True)
        return self._posterior_update(self._sum_losses(losses), None)

    def _sum_losses(self, losses, divisor=1):
        """svgd.py:66,72: ``total_loss = tensor(0.0); total_loss += loss`` per particle -- the same fp32 sum in the same
        order (0 + x == x bit for bit, so the first loss starts it), by ONE launch (bde_mean_scalars) when the losses are
        fp32 scalars on the particles' device, by torch's adds otherwise (a half-precision or off-device loss).  A small
        model's step is launch-bound: seven torch adds cost more host time than its whole posterior update.  ``divisor``:
        svgd.py:105's ``/ particle_count`` in the same launch, rounded like torch's GPU kernel for it (sum * fl(1 / count)).
        With the native host helper the checks and the call are one C++ function (host.cpp mean_losses).  The launch is
        taken only behind its device-verification gate ("mean_scalars"); torch's adds and division are the default."""
        dev = self._P.device if self._P is not None else self._Pown.device
        n = len(losses)
        if 1 < n <= 64 and hasattr(self._ops, "mean_scalars") and self._gate("mean_scalars"):
            native = self._native_mean_losses()
            stream = _raw_stream_of(dev) if native is not None else -1
            if stream >= 0:
                total = torch.empty((), dtype=torch.float32, device=dev)
                if native[0](losses, total, float(divisor), native[1], stream):
                    return total
            elif all(t.dtype == torch.float32 and t.numel() == 1 and t.device == dev for t in losses):
                total = torch.empty((), dtype=torch.float32, device=dev)
                self._ops.mean_scalars(losses, total, float(divisor))
                return total
        total = losses[0].to(device=dev, dtype=torch.float32, copy=True)
        for t in losses[1:]:
            total += t
        return total if divisor == 1 else total / divisor

    def _native_mean_losses(self):
        """(host.cpp mean_losses, address of this backend's bde_mean_scalars), or None: no host helper / a backend without a
        C entry point (the tests' checker)."""
        if self._mean_losses is None:
            self._mean_losses = False
            entry = getattr(self._ops, "mean_scalars_entry", None)
            if entry is not None:
                from . import _host
                native = _host.load()
                if native is not None and hasattr(native, "mean_losses"):
                    self._mean_losses = (native.mean_losses, int(entry()))
        return self._mean_losses or None

    def _particle_set(self):
        """The native object that runs the per-particle loops over its own tensor lists (csrc/host.cpp ParticleSet);
        None without the host helper.  Rebuilt whenever the particle views change (dimension sharding)."""
        if self._pset is None:
            from . import _host
            native = _host.load()
            if native is None or not hasattr(native, "ParticleSet"):
                self._pset = False
            else:
                self._pset = native.ParticleSet(self._plist, [v if v is not None else [] for v in self._pviews],
                                                [v if v is not None else [] for v in self._gviews])
        return self._pset or None

    def _begin_particle(self, particle_idx: int) -> None:
        """_use_particle (svgd.py:120-127) and base_optimizer.zero_grad() (svgd.py:70): the parameters view particle
        ``particle_idx`` and carry no gradient, so backward() hands over the fresh tensors its kernels produce."""
        pset = self._particle_set()
        if pset is not None:
            pset.begin(particle_idx)
            return
        repoint(self._plist, self._pviews[particle_idx], None)
        clear_grads(self._plist)

    def _end_particle(self, particle_idx: int) -> None:
        """_store_grads (svgd.py:129-133) without its clones: the tensors autograd just produced are kept BY REFERENCE and
        their addresses recorded (a gradient the kernels cannot read in place -- missing, strided, unaligned -- is
        copied / zeroed into the particle's flat gradient row instead, and that address recorded).  The update kernels
        read the gradients where they are; nothing is copied on the single-GPU path."""
        if self._seg is None:
            adopt_grads(self._plist, self._gviews[particle_idx])
            return
        if self._ov is not None:
            self._check_overlap_grads()
        if self._seg_host is None:
            self._seg_host = self._seg.staging()
        if self._exchange == "alltoall":
            particle_row = particle_idx - self._local_particles().start
        else:
            particle_row = particle_idx
        pset = self._particle_set()
        if pset is not None:
            pset.end(particle_idx, self._seg_host, particle_row, self._seg.m, self._zeros.data_ptr())
        else:
            self._retained[particle_row] = collect_grads(self._plist, self._gviews[particle_idx], self._seg_host,
                                                         particle_row, self._seg.m, self._zeros.data_ptr())
            clear_grads(self._plist)
        if self._ov is not None:
            self._launch_ready_chunks(force=True)         # tensors without a gradient: their chunks leave now

    def _check_overlap_grads(self) -> None:
        """overlap_backward: the chunks of the last local particle left while its backward pass was still running, each
        carrying the gradients as they were when autograd produced them.  A closure that touches the gradients AFTER
        ``backward()`` -- clip_grad_norm_, manual scaling, a second backward that accumulates -- would change them
        behind the exchange (the earlier local particles, collected after their closures, WOULD carry such edits): that
        is refused loudly instead of producing an inconsistent -phi.  A rank-local silent redo is not an option: the
        ranks must issue the same sequence of collectives."""
        for k, sent in self._ov["seen"].items():
            g = self._plist[k].grad
            now = (g.data_ptr(), g._version) if g is not None else None
            if now != sent:
                for work in self._ov["works"]:
                    if work is not None:
                        work.wait()                  # nothing of this step stays in flight behind the exception
                self._ov = None
                raise RuntimeError(
                    f"SVGDOptimizer(overlap_backward=True): the gradient of parameter {k} changed after its column chunk "
                    "had been sent (in-place edit, clipping, or a second backward pass inside backward_closure).  With "
                    "overlap_backward the closure must call backward() exactly once and leave the gradients alone "
                    "afterwards; use overlap_backward=False otherwise")

    def _grads_to_rows(self, G: torch.Tensor, row0: int, n_rows: int) -> None:
        """The gradients recorded by _end_particle packed into rows [row0, row0 + n_rows) of the flat buffer ``G`` (ONE
        launch for all of them): what a collective sends and what the small-model kernel reads.  No-op when the caller
        wrote the flat rows itself."""
        if self._seg_host is None:
            return
        self._seg.upload()
        self._seg_host = None
        self._ops.svgd_gather_seg(G, self._seg, row0, n_rows)
        self._release_grads()

    def _take_segments(self):
        """The segment table of this step, uploaded, if step() assembled one (then the kernels read the gradients where
        autograd left them); None when the flat gradient rows hold the gradients."""
        if self._seg_host is None:
            return None
        self._seg.upload()
        self._seg_host = None
        return self._seg

    def _release_grads(self) -> None:
        """After the update has been enqueued the gradient tensors may go back to the allocator (stream-ordered reuse)."""
        self._retained = [None] * len(self._retained)
        if self._pset:
            self._pset.release()

    def _posterior_update(self, total_loss, grad_scaler=None, mean_taken=False):
        """Everything after the forward/backward passes (svgd.py:82-105): gradient exchange (multi-GPU), kernel
        statistics, -phi and the base-optimizer applications.  ``total_loss`` = sum of this rank's particle losses
        (``mean_taken``: already divided by the particle count, single-GPU fast path); returns the mean loss over all
        particles.  (bench.py times exactly this.)"""
        base = self.state["__base_optimizer"]
        m = self.state["__particle_count"]
        with torch.no_grad():
            fused = (self._fuse or self._fuse_staged) and (grad_scaler is None or not grad_scaler.is_enabled())
            if self._fused_decision is None:
                self._fused_decision = fused
            elif self._fused_decision != fused:
                raise RuntimeError("fuse_base_optimizer: the GradScaler was enabled/disabled mid-run; the fused path keeps "
                                   "the shared optimizer state in its own flat buffers, so the choice must not change")
            if self._exchange == "alltoall":
                total_loss = self._step_dimension_sharded(total_loss, base)
            elif self._chunks is not None:
                total_loss = self._step_pipelined(total_loss, base, fused, grad_scaler)
            else:
                total_loss = self._step_replicated(total_loss, base, fused and self._fuse, grad_scaler,
                                                   staged_apply=fused and self._fuse_staged)
        return total_loss if mean_taken else total_loss / m

    # ---- replicated particles: one gather (or none), then the update ----------------------------------------
    def _stat_args(self):
        return (float(self.state["__l2_reg"]), float(self.state["__kernel_grad_scale"]),
                float(self.state["__dataset_size"]), -1.0)

    def _step_replicated(self, total_loss, base, fused, grad_scaler, staged_apply=False):
        m, d = self.state["__particle_count"], self._layout.d
        pending = None
        if self._sharded:
            per = m // self._world
            self._grads_to_rows(self._G, self._rank * per, per)          # own rows packed for the collective: one launch
            pending = self._start_gradient_exchange(total_loss)
        elif self._small_model(m, d):
            self._step_small_model(base, fused, grad_scaler, m, d)
            return total_loss
        # The Gram pass and the kernel statistics need only the (replicated) particles: they run while the
        # gradient all-gather is in flight.  (Skipped when the previous fused kernel already left the Gram.)
        if m > 16:
            self._grads_to_rows(self._G, 0, m)                           # the blocked kernels read flat rows
        if not (fused and self._reuse_gram and self._gram_valid):
            self._ops.svgd_gram(self._P, d, self._ws)
        self._ops.svgd_kstats(self._ws, m, *self._stat_args(), self._kstat)
        if pending is not None:
            total_loss = self._finish_gradient_exchange(pending)
        seg = self._take_segments()                                      # None: the gradients are in the flat rows
        if fused:
            # ONE pass: -phi in registers, M shared-state optimizer applications, updated particles out
            self._fused_apply(base, [(self._P, self._G, d, 0)], ws_next=self._ws if self._reuse_gram else None, seg=seg)
            self._gram_valid = self._reuse_gram
            self._use_particle(m - 1)    # the reference leaves the model aliased to the last particle
        else:
            self._gram_valid = False
            # svgd.py:86-89: -phi overwrites the gradient rows
            if seg is not None:
                self._ops.svgd_combine_seg(self._P, seg, self._G, d, self._kstat)
            elif m <= 16:
                self._ops.svgd_combine(self._P, self._G, self._G, d, self._kstat)
            else:
                # the blocked path for > 16 particles produces 16 rows per pass and re-reads all of G
                if self._tmp is None:
                    self._tmp = torch.zeros_like(self._G)
                self._ops.svgd_combine(self._P, self._G, self._tmp, d, self._kstat)
                if not staged_apply:
                    self._G.copy_(self._tmp)
            if staged_apply:
                # > 16 particles with a fusable base optimizer: its particle_count applications in ONE launch
                self._fused_apply(base, [(self._P, self._tmp if m > 16 else self._G, d, 0)], staged=True)
                self._use_particle(m - 1)
            else:
                self._apply_base_optimizer(base, grad_scaler)
        self._release_grads()
        return total_loss

    # ---- small models on one GPU: the whole update by the small-model kernel (two launches) -------------------
    def _small_model(self, m, d) -> bool:
        """Does this optimizer update with the small-model kernel?  ``single_launch="two"`` asks for it, ``None`` takes it
        once its device-verification record exists (see the class docstring); either way only where the kernel supports the
        problem on this device (a property of M, D and the device: asked once)."""
        if self._small_ok is None:
            want = self._single_launch == "two" or (self._single_launch is None and self._gate("svgd_small"))
            self._small_ok = bool(want) and bool(self._ops.svgd_small_supported(m, d))
        return self._small_ok

    def _step_small_model(self, base, fused, grad_scaler, m, d) -> None:
        """ONE default and ONE fallback per case.  Fused base optimizer: statistics, -phi and the M shared-state optimizer
        applications by bde_svgd_step_small_sgd / _adam -- issued together with the gradient packing by one native call
        (host.cpp small_step_*) behind its gate, by the two Python wrappers otherwise (same C-ABI calls, same arguments);
        graph_replay=True (opt-in) replays a recording of exactly those launches.  Any other base optimizer: -phi by
        bde_svgd_step_small, then the torch loop."""
        self._gram_valid = False
        if fused:
            if not (self._graph_replay and self._replay_small_sgd(base, m, d)) and \
                    not (self._gate("small_step_host") and self._native_small_step(base, m, d)):
                self._grads_to_rows(self._G, 0, m)                       # this kernel reads flat rows
                self._fused_apply(base, [(self._P, self._G, d, 0)], single_launch=True)
            self._use_particle(m - 1)
            return
        self._grads_to_rows(self._G, 0, m)
        # svgd.py:86-89: -phi overwrites the gradient rows
        self._ops.svgd_step_small(self._P, self._G, self._G, d, *self._stat_args(), self._ws, self._kstat)
        self._apply_base_optimizer(base, grad_scaler)
        self._release_grads()

    def _replay_small_sgd(self, base, m, d) -> bool:
        """``graph_replay=True``: a small model's step on one GPU is host-bound -- ~15 us of kernels behind four launches
        (segment-table upload, gradient packing, the update's two launches).  With torch.optim.SGD as the base optimizer
        every scalar those launches take is constant between LR-scheduler steps (the reference's CIFAR loop steps its
        scheduler once per epoch, experiments/cifar/cifar.py:169-172), so the sequence is recorded once per (staging slot of
        the table, scalars) in a hipGraph and replayed: one launch per step.  Same kernels, same arguments, same order -- the
        results are those of the eager path bit for bit.  Returns False (caller runs the eager path) for another base
        optimizer, without a segment table of this step, on the first steps (the library's lazy initialisation must not
        run inside a capture; the momentum buffers' first step has scalars of its own) and when the scalars change too
        often for a recording to pay (a per-STEP scheduler)."""
        if (not isinstance(base, torch.optim.SGD) or self._seg_host is None or not self._graphs_possible()
                or self._graph_captures > 8 + self._graph_replays // 64):
            return False
        g0 = self._fused_hyper(base)
        if g0.get("maximize", False):
            raise RuntimeError("fuse_base_optimizer: maximize=True is not supported")
        st = self._fused_buffers(base, "sgd")
        if st["first"] or self._graph_eager_steps < 3:
            self._graph_eager_steps += 1
            return False
        seg = self._seg
        slot = seg._slot
        l2, scale, n, _ = self._stat_args()
        scalars = (float(g0["lr"]), float(g0["momentum"]), float(g0["dampening"]), float(g0["weight_decay"]),
                   bool(g0["nesterov"]), l2, scale, n)
        # a recording holds the ADDRESSES it was made with: the key carries them, so a buffer that was re-allocated
        # (load_state_dict builds a new momentum buffer, set_particles under dimension sharding new rows) can never
        # meet a recording of its predecessor
        key = (slot, scalars, st["buf"].data_ptr(), self._P.data_ptr(), self._G.data_ptr(), self._ws.data_ptr(),
               self._kstat.data_ptr(), seg.ptrs.data_ptr())
        graph = self._graphs.get(key)
        if graph is None:
            graph = self._record_small_sgd(slot, scalars, st["buf"], m, d)
            if graph is None:
                return False                                             # the eager path performs this step
            if len(self._graphs) >= 12:                                 # (a handful of learning rates x three slots)
                self._graphs.clear()
            self._graphs[key] = graph
            self._graph_captures += 1
        graph.replay()
        self._graph_replays += 1
        seg.uploaded()                                                   # what SegTable.upload does around its copy
        self._seg_host = None
        self._release_grads()
        self._fused_advance()
        return True

    def _record_small_sgd(self, slot, scalars, buf, m, d):
        """One recording of _small_sgd_launches, or None: a capture that fails -- another thread made a HIP call the capture
        mode forbids, the runtime refused a node -- switches graph replay OFF for this optimizer instead of aborting the
        training step (nothing was enqueued: the caller's eager path runs the step).  ``thread_local``: only THIS thread's
        calls are policed, so a DataLoader's pinning thread cannot invalidate the capture."""
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                self._small_sgd_launches(slot, scalars, buf, m, d)
        except Exception as e:                                            # noqa: BLE001 -- whatever the runtime raised
            import warnings
            warnings.warn(f"SVGDOptimizer(graph_replay=True): recording the step failed ({type(e).__name__}: {e}); "
                          "continuing with eager launches")
            self._drop_graphs()
            self._graph_replay = False
            return None
        return graph

    def _drop_graphs(self) -> None:
        """Forget every recording (their buffers are about to change) and start over with eager steps."""
        self._graphs.clear()
        self._graph_eager_steps = 0

    def _native_small_step(self, base, m, d) -> bool:
        """The small-model update of a step whose gradients sit in the segment table -- upload, bde_svgd_gather_seg,
        bde_svgd_step_small_sgd / _adam -- with both C-ABI calls issued by ONE native function (host.cpp small_step_*)
        instead of two Python wrappers: the step is host-bound, and this is what _grads_to_rows + _fused_apply(single_launch)
        enqueue, argument for argument.  False: no host helper / table / plain SGD or Adam -- the caller takes that path."""
        if self._seg_host is None:
            return False
        native = self._small_step_native
        if native is None:
            native = False
            if hasattr(self._ops, "entry"):
                from . import _host
                mod = _host.load()
                if mod is not None and hasattr(mod, "small_step_sgd"):
                    native = (mod, self._ops.entry("bde_svgd_gather_seg"), self._ops.entry("bde_svgd_step_small_sgd"),
                              self._ops.entry("bde_svgd_step_small_adam"))
            self._small_step_native = native
        if native is False:
            return False
        sgd = isinstance(base, torch.optim.SGD)
        if not sgd and type(base) is not torch.optim.Adam:
            return False
        g0 = self._fused_hyper(base)
        if g0.get("maximize", False) or (not sgd and (g0.get("amsgrad", False) or g0.get("decoupled_weight_decay", False))):
            return False                                                 # (_fused_apply raises, naming the option)
        stream = _raw_stream_of(self._P.device)
        if stream < 0:
            return False
        mod, e_gather, e_sgd, e_adam = native
        st = self._fused_buffers(base, "sgd" if sgd else "adam")
        seg = self._seg
        seg.upload()
        self._seg_host = None
        l2, scale, n, _ = self._stat_args()
        if sgd:
            mod.small_step_sgd(e_gather, e_sgd, seg.ptrs, seg.chunks, self._P, self._G, st["buf"], self._ws, self._kstat, d, l2,
                               scale, n, float(g0["lr"]), g0["momentum"], g0["dampening"], g0["weight_decay"], bool(g0["nesterov"]),
                               bool(st["first"]), stream)
        else:
            mod.small_step_adam(e_gather, e_adam, seg.ptrs, seg.chunks, self._P, self._G, st["exp_avg"], st["exp_avg_sq"],
                                self._ws, self._kstat, d, l2, scale, n, float(g0["lr"]), g0["betas"][0], g0["betas"][1],
                                g0["eps"], g0["weight_decay"], int(st["step"]), stream)
        self._release_grads()
        self._fused_advance()
        return True

    def _graphs_possible(self) -> bool:
        return self._P.is_cuda

    def _small_sgd_launches(self, slot, scalars, buf, m, d) -> None:
        """What _grads_to_rows + _fused_apply(single_launch=True) enqueue for an SGD base optimizer after its first step."""
        seg = self._seg
        seg.ptrs.copy_(seg.host[slot], non_blocking=True)               # (read from the pinned table at every replay)
        self._ops.svgd_gather_seg(self._G, seg, 0, m)
        self._ops.svgd_step_small_sgd(self._P, self._G, buf, d, *scalars[5:], self._ws, self._kstat, *scalars[:5], False)

    def _apply_base_optimizer(self, base, grad_scaler):
        """svgd.py:92-103: hand row i of -phi to the base optimizer as the gradient of particle i, for every i."""
        OptState = _opt_state()
        pset = self._particle_set()
        for particle_idx in range(self.state["__particle_count"]):
            if pset is not None:
                pset.set_grads(particle_idx)
            else:
                repoint(self._plist, self._pviews[particle_idx], self._gviews[particle_idx])
            if grad_scaler is not None:
                self._set_grad_scaler_state(grad_scaler, OptState.UNSCALED, base)
                grad_scaler.step(base)
            else:
                base.step()

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

    # ---- replicated particles, pipelined: gather chunk c+1 while chunk c is being updated --------------------
    def _step_pipelined(self, total_loss, base, fused, grad_scaler):
        import torch.distributed as dist
        m, d = self.state["__particle_count"], self._layout.d
        per = m // self._world
        lo = self._rank * per
        if self._ov is not None:
            # the chunks left while the last backward pass was still running (_grad_ready / _end_particle)
            works, self._ov = self._ov["works"], None
            self._seg.upload()                               # closes the step's staging slot
            self._seg_host = None
            self._release_grads()
        else:
            self._grads_to_rows(self._G, lo, per)                        # own rows packed for the collective: one launch
            self._G[lo:lo + per, d] = total_loss / per
            works = []
            for (c0, c1), stage in zip(self._chunks, self._stage):
                # a row chunk is contiguous; several own rows are packed first (1/W of the data)
                send = self._G[lo, c0:c1] if per == 1 else self._G[lo:lo + per, c0:c1].contiguous().view(-1)
                works.append(dist.all_gather_into_tensor(stage.view(-1), send, group=self._pg, async_op=True))
        self._ops.svgd_gram(self._P, d, self._ws)            # needs the particles only: hidden behind the exchange
        self._ops.svgd_kstats(self._ws, m, *self._stat_args(), self._kstat)
        self._gram_valid = False
        for (c0, c1), stage, work in zip(self._chunks, self._stage, works):
            work.wait()                                       # the compute stream waits for THIS chunk only
            if c0 <= d < c1:
                total_loss = stage[:, d - c0].sum()
            dc = min(d, c1) - c0
            if dc <= 0:
                continue
            if fused:
                self._fused_apply(base, [(self._P[:, c0:c1], stage, dc, c0)], ws_next=None, advance=False)
            else:
                self._ops.svgd_combine(self._P[:, c0:c1], stage, self._G[:, c0:c1], dc, self._kstat)
        self._G[lo:lo + per, d] = 0
        if fused:
            self._fused_advance()
            self._use_particle(m - 1)
        else:
            self._apply_base_optimizer(base, grad_scaler)
        return total_loss

    # ---- dimension-sharded particles (exchange="alltoall") ---------------------------------------------------
    def _init_dimension_sharding(self):
        """Split the (broadcast, identical) particle matrix: this rank keeps the full rows of its own particles
        (for forward/backward) and its column slice of ALL particles (for the update)."""
        if not self._fuse:
            raise ValueError("exchange='alltoall' keeps the shared base-optimizer state sharded by columns and therefore "
                             "needs fuse_base_optimizer=True (torch.optim.SGD or Adam) and particle_count <= 16")
        m, d, ld, dev = self.state["__particle_count"], self._layout.d, self._layout.ld, self._P.device
        w, r = self._world, self._rank
        per = m // w
        lo = r * per
        sl = pad4((ld + w - 1) // w)
        self._sl, self._ldw = sl, sl * w
        full = torch.zeros((m, self._ldw), dtype=torch.float32, device=dev)
        full[:, :ld] = self._P
        self._Pown = full[lo:lo + per].clone()
        self._Gown = torch.zeros_like(self._Pown)
        self._Ps = full[:, r * sl:(r + 1) * sl].clone()
        self._Gs = torch.zeros_like(self._Ps)
        self._slice_d = max(0, min(d, (r + 1) * sl) - r * sl)          # valid columns of this rank's slice
        self._msg = torch.zeros(self._ops.GMAT_DOUBLES + 1, dtype=torch.float64, device=dev)
        self._msgs = torch.zeros((w, self._ops.GMAT_DOUBLES + 1), dtype=torch.float64, device=dev)
        self._reuse_gram = False
        del self._P, self._G, full
        self._P = self._G = None
        for i in range(m):
            if lo <= i < lo + per:
                self._pviews[i] = self._layout.views(self._Pown[i - lo])
                self._gviews[i] = self._layout.views(self._Gown[i - lo])
                for param, view in zip(self._plist, self._pviews[i]):
                    self.state[param][f"particle_{i}"] = view
            else:
                self._pviews[i] = self._gviews[i] = None
                for param in self._plist:
                    self.state[param].pop(f"particle_{i}", None)
        self._pset = None                                          # the views changed: the native loops get the new lists
        self._use_particle(lo)

    def _all_to_all(self, out: torch.Tensor, inp: torch.Tensor):
        """out[s] <- rank s's inp[this rank]; both [W, per, sl] contiguous.  RCCL: one all_to_all_single.  Backends
        without a device all-to-all (gloo in the one-GPU test harness) gather everything and select."""
        import torch.distributed as dist
        if dist.get_backend(self._pg) == "nccl":
            return dist.all_to_all_single(out.view(-1), inp.view(-1), group=self._pg, async_op=True)
        everything = torch.empty((self._world,) + tuple(inp.shape), dtype=inp.dtype, device=inp.device)
        dist.all_gather_into_tensor(everything.view(-1), inp.reshape(-1).clone(), group=self._pg)
        out.copy_(everything[:, self._rank])
        return None

    def _step_dimension_sharded(self, local_loss_sum, base):
        import torch.distributed as dist
        m, w, r, sl = self.state["__particle_count"], self._world, self._rank, self._sl
        per = m // w
        self._grads_to_rows(self._Gown, 0, per)                          # own rows packed for the all-to-all: one launch
        # (1) the slice's Gram block + this rank's loss: one tiny all-gather, issued first
        if self._slice_d > 0:
            self._ops.svgd_gram(self._Ps, self._slice_d, self._ws)
            self._ops.svgd_gram_finish(self._ws, m, self._msg)
        else:
            self._msg.zero_()
            self._msg[256] = 8.0 if m <= 8 else 16.0
        self._msg[-1] = local_loss_sum.double()
        stats = dist.all_gather_into_tensor(self._msgs.view(-1), self._msg, group=self._pg, async_op=True)
        # (2) gradient rows -> gradient slices
        if per == 1:
            send = self._Gown.view(w, 1, sl)
        else:
            send = self._Gown.view(per, w, sl).transpose(0, 1).contiguous()
        grads = self._all_to_all(self._Gs.view(w, per, sl), send)
        stats.wait()
        self._ops.svgd_kstats_gmat(self._msgs, m, *self._stat_args(), self._kstat)
        total_loss = self._msgs[:, -1].sum().float()
        if grads is not None:
            grads.wait()
        # (3) the fused update of this rank's slice of all particles (shared optimizer state sharded the same way)
        if self._slice_d > 0:
            self._fused_apply(base, [(self._Ps, self._Gs, self._slice_d, 0)], ws_next=None)
        else:
            self._fused_apply(base, [], ws_next=None)
        # (4) updated slices -> the particles' owners
        if per == 1:
            back = self._all_to_all(self._Pown.view(w, 1, sl), self._Ps.view(w, 1, sl))
            if back is not None:
                back.wait()
        else:
            tmp = torch.empty((w, per, sl), dtype=torch.float32, device=self._Ps.device)
            back = self._all_to_all(tmp, self._Ps.view(w, per, sl))
            if back is not None:
                back.wait()
            self._Pown.view(per, w, sl).copy_(tmp.transpose(0, 1))
        self._use_particle(r * per + per - 1)
        return total_loss

    # ---- fused shared-state optimizer -----------------------------------------------------------------------
    def _fused_hyper(self, base):
        groups = base.param_groups
        g0 = groups[0]
        keys = [k for k in g0 if k != "params"]
        for g in groups[1:]:
            if any(g[k] != g0[k] for k in keys):
                raise RuntimeError("fuse_base_optimizer needs identical hyper-parameters in all param groups (they were "
                                   "equal when this SVGDOptimizer was constructed): give the groups one set of "
                                   "hyper-parameters or construct with fuse_base_optimizer=False")
        # fusability was decided at construction; the fused kernels never call base.step(), so a step hook registered on
        # the base optimizer SINCE would silently never run -- refuse instead
        if getattr(base, "_optimizer_step_pre_hooks", None) or getattr(base, "_optimizer_step_post_hooks", None):
            raise RuntimeError("the base optimizer has step hooks, which the fused SVGD update (fuse_base_optimizer) cannot "
                               "call: register hooks before constructing the SVGDOptimizer (then 'auto' keeps the torch "
                               "loop) or pass fuse_base_optimizer=False")
        return g0

    def _fused_buffers(self, base, kind):
        """The SHARED optimizer state as flat buffers (one element per parameter / per slice column).  Replicated
        layouts alias it into ``base.state[param]`` (``momentum_buffer`` / ``exp_avg`` / ``exp_avg_sq`` views and
        ``step``), so ``base.state_dict()`` is complete and a run can be resumed unfused or in the reference."""
        if self._fused_state is not None:
            return self._fused_state
        dev = self._params_device()
        loaded, self._fused_loaded = getattr(self, "_fused_loaded", None), None
        sharded = self._exchange == "alltoall"
        n = self._sl if sharded else self._layout.ld
        if kind == "sgd":
            st = {"kind": "sgd", "buf": torch.zeros(n, device=dev), "first": True}
            names = {"buf": "momentum_buffer"}
        else:
            st = {"kind": "adam", "exp_avg": torch.zeros(n, device=dev), "exp_avg_sq": torch.zeros(n, device=dev),
                  "step": 0}
            names = {"exp_avg": "exp_avg", "exp_avg_sq": "exp_avg_sq"}
        # What the state starts from, as FULL-length flat vectors: a checkpoint written by the fused path (its "__fused"
        # entry: full-length buffers in every exchange mode), else whatever the base optimizer already holds per tensor
        # (a resumed unfused run, or a checkpoint written by the reference: momentum_buffer / exp_avg / exp_avg_sq)
        ld = self._layout.ld
        full = {}
        if loaded is not None and loaded.get("kind") == kind:
            for key in names:
                if torch.is_tensor(loaded.get(key)) and loaded[key].numel() in (n, ld):
                    full[key] = loaded[key].to(dev)
            for key in ("first", "step"):
                if key in loaded:
                    st[key] = loaded[key]
        else:
            seen = False
            for key, name in names.items():
                vec = torch.zeros(ld, device=dev)
                for v, p in zip(self._layout.views(vec), self._plist):
                    old = base.state.get(p, {}).get(name)
                    if torch.is_tensor(old):
                        v.copy_(old.to(dev))
                        seen = True
                full[key] = vec
            if not seen:
                full = {}
            for p in self._plist:
                if kind == "adam" and "step" in base.state.get(p, {}):
                    st["step"] = int(base.state[p]["step"])
            if kind == "sgd" and seen:
                st["first"] = False
        for key, vec in full.items():
            if vec.numel() == n:
                st[key].copy_(vec)
            else:                                  # a full-length vector into this rank's column slice
                lo = self._rank * self._sl
                take = max(0, min(ld, lo + n) - lo)
                st[key][:take].copy_(vec[lo:lo + take])
        if not sharded:
            # ... and publish the flat buffers as that state
            views = {key: self._layout.views(st[key]) for key in names}
            if kind == "adam":
                # torch's Adam keeps one step tensor per parameter; they are refreshed from the shared counter
                # whenever the base optimizer's state_dict is taken
                st["step_tensors"] = [torch.tensor(float(st["step"])) for _ in self._plist]

                def refresh_steps(_optimizer, st=st):
                    for t in st["step_tensors"]:
                        t.fill_(float(st["step"]))
                if hasattr(base, "register_state_dict_pre_hook"):
                    base.register_state_dict_pre_hook(refresh_steps)
            for i, p in enumerate(self._plist):
                entry = base.state[p]
                for key, name in names.items():
                    entry[name] = views[key][i]
                if kind == "adam":
                    entry["step"] = st["step_tensors"][i]
        self._fused_state = self.state["__fused"] = st
        return st

    def _fused_apply(self, base, pieces, ws_next=None, advance=True, single_launch=False, seg=None, staged=False) -> None:
        """-phi and the M sequential base-optimizer applications with shared state in ONE kernel per piece
        (bde_svgd_fused_sgd / bde_svgd_fused_adam); ``pieces`` = (P, G, valid columns, column offset into the state
        buffers).  Hyper-parameters are read from the base optimizer's param_groups every step, so LR schedulers
        keep working."""
        g0 = self._fused_hyper(base)
        if isinstance(base, torch.optim.SGD):
            if g0.get("maximize", False):
                raise RuntimeError("fuse_base_optimizer: maximize=True is not supported")
            st = self._fused_buffers(base, "sgd")
            for P, G, d, c0 in pieces:
                if staged:                           # G holds -phi already (blocked update): the applications alone
                    self._ops.svgd_apply_sgd(P, G, st["buf"][c0:], d, g0["lr"], g0["momentum"], g0["dampening"],
                                             g0["weight_decay"], g0["nesterov"], st["first"])
                    continue
                if single_launch:
                    l2, scale, n, _ = self._stat_args()
                    args = (P, G, st["buf"], d, l2, scale, n, self._ws, self._kstat, g0["lr"], g0["momentum"],
                            g0["dampening"], g0["weight_decay"], g0["nesterov"], st["first"])
                    self._ops.svgd_step_small_sgd(*args)
                    continue
                if seg is not None:
                    self._ops.svgd_fused_sgd_seg(P, seg, st["buf"], d, self._kstat, g0["lr"], g0["momentum"],
                                                 g0["dampening"], g0["weight_decay"], g0["nesterov"], st["first"],
                                                 ws_next=ws_next)
                    continue
                self._ops.svgd_fused_sgd(P, G, st["buf"][c0:], d, self._kstat, g0["lr"], g0["momentum"], g0["dampening"],
                                         g0["weight_decay"], g0["nesterov"], st["first"], ws_next=ws_next)
        elif type(base) is torch.optim.Adam:
            if g0.get("amsgrad", False) or g0.get("maximize", False) or g0.get("decoupled_weight_decay", False):
                # the kernels apply torch.optim.Adam's COUPLED decay (g += wd * theta); AdamW-style decay would be a
                # different update, so it is refused rather than silently replaced (fuse_base_optimizer="auto" keeps the
                # torch loop for such an optimizer)
                raise RuntimeError("fuse_base_optimizer: amsgrad / maximize / decoupled_weight_decay are not supported")
            st = self._fused_buffers(base, "adam")
            for P, G, d, c0 in pieces:
                if staged:
                    self._ops.svgd_apply_adam(P, G, st["exp_avg"][c0:], st["exp_avg_sq"][c0:], d, float(g0["lr"]),
                                              g0["betas"][0], g0["betas"][1], g0["eps"], g0["weight_decay"], st["step"])
                    continue
                if single_launch:
                    l2, scale, n, _ = self._stat_args()
                    args = (P, G, st["exp_avg"], st["exp_avg_sq"], d, l2, scale, n, self._ws, self._kstat, float(g0["lr"]),
                            g0["betas"][0], g0["betas"][1], g0["eps"], g0["weight_decay"], st["step"])
                    self._ops.svgd_step_small_adam(*args)
                    continue
                if seg is not None:
                    self._ops.svgd_fused_adam_seg(P, seg, st["exp_avg"], st["exp_avg_sq"], d, self._kstat, float(g0["lr"]),
                                                  g0["betas"][0], g0["betas"][1], g0["eps"], g0["weight_decay"],
                                                  st["step"], ws_next=ws_next)
                    continue
                self._ops.svgd_fused_adam(P, G, st["exp_avg"][c0:], st["exp_avg_sq"][c0:], d, self._kstat, float(g0["lr"]),
                                          g0["betas"][0], g0["betas"][1], g0["eps"], g0["weight_decay"], st["step"],
                                          ws_next=ws_next)
        else:
            raise RuntimeError(f"fuse_base_optimizer supports torch.optim.SGD and torch.optim.Adam, got {type(base)}")
        if advance:
            self._fused_advance()

    def _fused_advance(self):
        """One SVGD step = particle_count applications of the shared optimizer."""
        st = self._fused_state
        if st["kind"] == "sgd":
            st["first"] = False
        else:
            st["step"] += self.state["__particle_count"]
        # the base optimizer HAS stepped (inside the kernel): keep torch's LR schedulers from warning that
        # scheduler.step() ran before optimizer.step() (they look at this flag, set by their wrapper of base.step)
        self.state["__base_optimizer"]._opt_called = True

    # ------------------------------------------------------------------
    def sample_parameters(self):
        '''Point the model at the next particle, round robin (svgd.py:107-112).  With dimension-sharded
        particles only this rank's own particles are available, so the cycle runs over those.'''
        count = self.state["__particle_count"]
        current = self.state["__current_particle"]
        if self._exchange == "alltoall":
            local = self._local_particles()
            self._use_particle(local.start + current % len(local))
        else:
            self._use_particle(current)
        self.state["__current_particle"] = (current + 1) % count

    def _params_for_particle(self, particle_idx):
        '''Particle i's tensors in parameter order (svgd.py:114-118).'''
        key = f"particle_{particle_idx}"
        return (self.state[param][key] for param in self._params())

    def _use_particle(self, particle_idx):
        '''Does *not* clone: updates of the model parameters are updates of the particle (svgd.py:120-127)'''
        pset = self._particle_set()
        if pset is not None:
            pset.use(particle_idx)
        else:
            repoint(self._plist, self._pviews[particle_idx], None)

    def get_base_optimizer(self):
        return self.state["__base_optimizer"]

    # ---- flat access (bench / multi-GPU tests / checkpoints) -----------
    def _particle_rows(self) -> torch.Tensor:
        """``[M, d]`` in the flat row layout (alignment padding included).  With exchange="alltoall" the slices are
        gathered first (a collective)."""
        d = self._layout.d
        if self._exchange != "alltoall":
            return self._P[:, :d]
        import torch.distributed as dist
        m, w, sl = self.state["__particle_count"], self._world, self._sl
        slices = torch.empty((w, m, sl), dtype=torch.float32, device=self._Ps.device)
        dist.all_gather_into_tensor(slices.view(-1), self._Ps.reshape(-1).clone(), group=self._pg)
        return slices.permute(1, 0, 2).reshape(m, w * sl)[:, :d]

    @property
    def particles(self) -> torch.Tensor:
        """The particles as ``[M, D]`` (the parameters' elements in parameter order): a view of the flat buffer when
        no tensor needed alignment padding, a gathered copy otherwise.  With exchange="alltoall" the slices are gathered
        first (a collective: every rank of the group must read this property)."""
        return self._layout.compact(self._particle_rows())

    def set_particles(self, particles: torch.Tensor) -> None:
        """Overwrite all particles with ``particles [M, D]`` (the layout ``.particles`` returns).  ``.particles`` itself is
        a copy whenever a tensor needed alignment padding, so in-place edits go through here.  Not available with
        exchange="alltoall" (load a checkpoint instead)."""
        if self._exchange == "alltoall":
            raise NotImplementedError("set_particles with exchange='alltoall': use load_state_dict")
        m = self.state["__particle_count"]
        if tuple(particles.shape) != (m, self._layout.n_valid):
            raise ValueError(f"expected a [{m}, {self._layout.n_valid}] tensor, got {tuple(particles.shape)}")
        with torch.no_grad():
            if self._layout.padded:
                self._P[:, self._layout.valid_index(self._P.device)] = particles.to(self._P.device, torch.float32)
            else:
                self._P[:, :self._layout.n_valid] = particles.to(self._P.device, torch.float32)
        self._gram_valid = False

    @property
    def kernel_stats(self) -> dict:
        m = self.state["__particle_count"]
        ks = self._kstat
        return {"kernel": ks[:m * m].view(m, m), "d2": ks[m * m:2 * m * m].view(m, m),
                "h": ks[2 * m * m + m], "median": ks[2 * m * m + m + 1]}

    def invalidate_gram(self) -> None:
        """Call after modifying the particles outside this optimizer when reuse_gram=True."""
        self._gram_valid = False

    def _gather_slices(self, vec: torch.Tensor) -> torch.Tensor:
        """This rank's ``[sl]`` slice of a column-sharded vector -> the full ``[ld]`` vector (a collective)."""
        import torch.distributed as dist
        out = torch.empty(self._world * self._sl, dtype=vec.dtype, device=vec.device)
        dist.all_gather_into_tensor(out, vec.contiguous().clone(), group=self._pg)
        return out[:self._layout.ld]

    def state_dict(self):
        """The reference's layout in every exchange mode (svgd.py:51-61 through ensemble.py:17-26): per-tensor
        ``particle_i`` entries for ALL particles and the pickled ``__base_optimizer``.  With exchange="alltoall" the
        particles and the shared optimizer state are sharded by columns, so this is a COLLECTIVE there (every rank of
        the group must call it; every rank gets the complete dict): the slices are gathered, the fused optimizer's state
        is published into ``base_optimizer.state`` in torch's own layout, and ``__fused`` carries full-length buffers."""
        if self._exchange != "alltoall":
            return super().state_dict()
        m, d, ld = self.state["__particle_count"], self._layout.d, self._layout.ld
        full = torch.zeros((m, ld), dtype=torch.float32, device=self._Ps.device)
        full[:, :d] = self._particle_rows()                                 # collective
        for i in range(m):
            for param, view in zip(self._plist, self._layout.views(full[i])):
                self.state[param][f"particle_{i}"] = view
        st = self._fused_state
        if st is not None:
            base = self.state["__base_optimizer"]
            names = {"buf": "momentum_buffer"} if st["kind"] == "sgd" else {"exp_avg": "exp_avg", "exp_avg_sq": "exp_avg_sq"}
            gathered = {key: self._gather_slices(st[key]) for key in names}     # collectives, same order on every rank
            published = dict(st, **gathered)
            has_state = st["kind"] == "adam" or not st["first"]
            for key, name in names.items():
                for p, v in zip(self._plist, self._layout.views(gathered[key])):
                    if has_state:
                        base.state[p][name] = v
            if st["kind"] == "adam":
                for p in self._plist:
                    base.state[p]["step"] = torch.tensor(float(st["step"]))
            self.state["__fused"] = published
        try:
            sd = super().state_dict()
            sd["state"] = {k: (dict(v) if isinstance(v, dict) else v) for k, v in sd["state"].items()}
        finally:
            self._realias_sharded_state()
            if st is not None:
                self.state["__fused"] = st
        return sd

    def _realias_sharded_state(self) -> None:
        """exchange="alltoall": ``state[param]["particle_i"]`` exists for this rank's own particles only (views of
        their rows)."""
        own = self._local_particles()
        for i in range(self.state["__particle_count"]):
            if i in own:
                for param, view in zip(self._plist, self._pviews[i]):
                    self.state[param][f"particle_{i}"] = view
            else:
                for param in self._plist:
                    self.state[param].pop(f"particle_{i}", None)

    def load_state_dict(self, state_dict):
        """Accepts the reference's layout (per-tensor ``particle_i`` entries): the
        values are copied into the flat buffer and the state re-aliased to it.  With exchange="alltoall" every rank
        loads the same (complete) dict and keeps its own particles' rows and its column slice of all particles and of
        the shared optimizer state; no communication."""
        super().load_state_dict(state_dict)
        self._keep_live_base_optimizer(self._live_base)
        self._gram_valid = False
        # shared optimizer state of the fused path: re-adopted (and re-published into base.state) at the next step
        self._fused_loaded = self.state.pop("__fused", None)
        self._fused_state = None
        self._drop_graphs()                                # recordings hold the old buffers' addresses
        self._fused_decision = None
        m = self.state["__particle_count"]
        if self._exchange == "alltoall":
            ld, sl, r = self._layout.ld, self._sl, self._rank
            own = self._local_particles()
            full = torch.zeros((m, self._ldw), dtype=torch.float32, device=self._Ps.device)
            with torch.no_grad():
                for i in range(m):
                    loaded = [self.state[param][f"particle_{i}"] for param in self._plist]
                    torch._foreach_copy_(self._layout.views(full[i, :ld]), [t.to(full.device) for t in loaded])
                self._Pown.copy_(full[own.start:own.stop])
                self._Ps.copy_(full[:, r * sl:(r + 1) * sl])
            self._realias_sharded_state()
            return
        with torch.no_grad():
            for i in range(m):
                for param, view in zip(self._plist, self._pviews[i]):
                    loaded = self.state[param][f"particle_{i}"]
                    if loaded.data_ptr() != view.data_ptr():
                        view.copy_(loaded)
                    self.state[param][f"particle_{i}"] = view
