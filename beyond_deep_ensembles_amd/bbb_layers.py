"""Mean-field Bayesian layers with the local reparameterisation trick, built on
``GaussianParameter`` (SURVEY.md section 8f, row 4 -- the callers of the BBB hot path).

Same constructor arguments, attributes (``weight`` / ``bias`` GaussianParameters,
``is_bayesian``, ``kl``) and forward semantics as the reference's ``BBBLinear``
(``src/algos/bbb_layers.py:10-90``) and ``BBBConv2d`` (``:105-160``) in their default
``sampling="activations"`` mode: the layer's pre-activations are sampled from
``N(x W_mu + b_mu, x^2 W_sigma^2 + b_sigma^2)`` (inputs^2 and variances clamped at 1e-4),
with ONE noise draw shared across the batch in eval mode when ``freeze_on_eval``.
``BBBLinear`` runs its whole forward as ONE fused op for batches of up to 128 rows (``bde_lrt_linear_fwd``: the
weights are streamed once, sigma^2 / x^2 are formed on the fly, both products run on the MFMA, the noise is applied
in the finish pass) and its backward as three launches.  ``BBBConv2d`` (round 4) runs both convolutions of
``bbb_layers.py:146-147`` as ONE dual-accumulator implicit GEMM with the sampling epilogue fused
(``bde_conv_lrt_fwd``), its backward as an input-gradient and a weight-gradient kernel of the same kind
(``bde_conv_lrt_bwd_*``); the weights are prepared once per version (``bde_conv_lrt_prep``).  Larger linear batches,
unsupported convolution geometries and eval-mode frozen noise keep the two stock GEMMs / convolutions, with every
element-wise piece around them fused: one pass per operand of the variance product (``bde_var_operand_*``) and one for
the epilogue (``bde_local_reparam_*``).  What BBBOptimizer needs from the layer -- mean / rho parameters paired through
GaussianParameter -- feeds the fused KL kernel.

Differences from the reference, on purpose:
* ``kl`` is evaluated lazily when read (the reference recomputes it on every forward although
  BBBOptimizer never reads it); ``BBBConv2d.kl`` uses the BIAS parameters for
  the bias term (the reference adds the weight KL twice).
* ``sampling="parameters"`` draws the weights with ``GaussianParameter.sample()`` (HIP draw
  kernel) and returns the plain layer output; the per-sample log-prob "kl" of the reference's
  variant (which calls an undefined helper) is not reproduced.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable
import torch.nn.functional as F

from . import _host
from .util import GaussianParameter, normal_like, _philox_stream

_CLAMP = 1e-4
# With the C++ autograd nodes of lib/_bde_host.so a fused element-wise pass costs about what one native ATen node costs,
# and every piece is fused at every size.  As a Python autograd Function a node costs 40-55 us of host time against ~8 us
# for a native one, so without the helper a piece gets its own pass only where the GPU time of the ATen sequence it
# replaces exceeds that (tools/conv_layer_bench.py: break-even at a few million elements).
_FUSE_MIN_ELEMS = 1 << 22


def _fuse_min(ops) -> int:
    return 0 if _native_nodes(ops) is not None else _FUSE_MIN_ELEMS



def _native_nodes(ops):
    """The C++ autograd nodes of lib/_bde_host.so (csrc/host_autograd.cpp) when the kernels are the HIP library's; the
    Python Functions below are the same nodes for any other backend (the tests' CPU checker) or without the helper."""
    from .ops import HipOps
    if not isinstance(ops, HipOps):
        return None
    mod = _host.load()
    return mod if mod is not None and hasattr(mod, "lrt_linear") else None


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


_LRT_TILE_ROWS = 128        # rows one bde_lrt_linear_fwd / _bwd launch takes (bde_lrt_linear_supported)


def _lrt_linear_tiled(x, w_mu, w_rho, b_mu, b_rho, clamp_bias, eps, seed, next_stream_id, ops, w_s2=None, w_ds2=None):
    """bbb_layers.py:61-80 for MORE than 128 rows with the kernels that exist (no new device code): the rows in
    tiles of 128, each tile one fused forward node (its own Philox stream id from ``next_stream_id()``: the activations'
    noise is i.i.d. per element either way, and a fixed tile size makes the draw a function of (seed, first stream id, row,
    column) alone), the outputs concatenated.  Backward: autograd runs the tiles' nodes in a fixed order and ACCUMULATES their
    weight / bias gradients in that order (each tile's bde_lrt_linear_bwd overwrites its own buffers; the sum over tiles is
    torch's add) -- bit-reproducible run to run; the input gradient of a tile is that tile's rows."""
    x2d = x.reshape(-1, x.shape[-1])
    e2d = None if eps is None else eps.reshape(x2d.shape[0], -1)
    outs = []
    for r0 in range(0, x2d.shape[0], _LRT_TILE_ROWS):
        r1 = min(x2d.shape[0], r0 + _LRT_TILE_ROWS)
        outs.append(_lrt_linear(x2d[r0:r1], w_mu, w_rho, b_mu, b_rho, clamp_bias, None if e2d is None else e2d[r0:r1], seed,
                                next_stream_id(), ops, w_s2, w_ds2))
    return torch.cat(outs, dim=0).view(x.shape[:-1] + (w_mu.shape[0],))


def _lrt_linear(x, w_mu, w_rho, b_mu, b_rho, clamp_bias, eps, seed, stream_id, ops, w_s2=None, w_ds2=None):
    native = _native_nodes(ops)
    if native is not None:
        return native.lrt_linear(x, w_mu, w_rho, b_mu, b_rho, clamp_bias, eps, seed, stream_id, w_s2, w_ds2)
    return _LrtLinear.apply(x, w_mu, w_rho, b_mu, b_rho, clamp_bias, eps, seed, stream_id, ops, w_s2, w_ds2)


class _SigmaCache:
    """sigma^2 = clamp(softplus(rho)^2, 1e-4) and its rho-derivative of a wide layer's weight matrix, computed ONCE per
    version of ``rho`` (bde_lrt_sigma_cache) and read by every fused forward / backward of that version: the weights
    only change at ``base_optimizer.step()``, but a BBB step runs ``mc_samples`` forward and backward passes
    (``bbb.py:63-67``).  Keyed on the parameter's storage address and autograd version counter, so any in-place change
    of rho -- an optimizer step, ``load_state_dict``, a manual edit -- refreshes it at the next forward.  A refresh
    writes NEW tensors: a forward whose backward has not run yet keeps the tensors of its own version.

    Writes through ``rho.data`` (``rho.data.fill_()``, a custom optimizer written the way the reference writes ``.data``)
    keep the address AND the version counter; for those the key also carries a process-wide epoch that
    ``BBBOptimizer.step`` advances when it begins and ends and ``BBBOptimizer.load_state_dict`` advances once
    (``invalidate_sigma_caches``), and a layer can drop its own cache (``layer.invalidate_sigma_cache()``): code that edits
    ``rho.data`` BETWEEN two forward passes of one step has to call one of the two."""

    epoch = 0

    def __init__(self):
        self.key = None
        self.s2 = self.ds2 = None

    def drop(self) -> None:
        self.key = None

    def get(self, rho: torch.Tensor, ops):
        key = (rho.data_ptr(), rho._version, tuple(rho.shape), _SigmaCache.epoch)
        if key != self.key:
            r = rho.detach().contiguous()
            s2, ds2 = torch.empty_like(r), torch.empty_like(r)
            ops.lrt_sigma_cache(r, s2, ds2)
            self.key, self.s2, self.ds2 = key, s2, ds2
        return self.s2, self.ds2


def invalidate_sigma_caches() -> None:
    """Every layer's cached sigma^2 is recomputed at its next forward (cheap: one pass per wide layer)."""
    _SigmaCache.epoch += 1


def _local_reparam(mean, var, eps, seed, stream_id, ops):
    native = _native_nodes(ops)
    if native is not None:
        return native.local_reparam(mean, var, eps, seed, stream_id)
    return _LocalReparam.apply(mean, var, eps, seed, stream_id, ops)


def _var_operand(v, mode, ops):
    native = _native_nodes(ops)
    if native is not None:
        return native.var_operand(v, mode)
    return _VarOperand.apply(v, mode, ops)


class _LocalReparam(torch.autograd.Function):
    """out = mean + sqrt(var) * eps with the fused HIP epilogue (bde_local_reparam_fwd/bwd)."""

    @staticmethod
    def forward(ctx, mean, var, eps, seed, stream_id, ops):
        m, v = mean.contiguous().view(-1), var.contiguous().view(-1)
        e = None if eps is None else eps.contiguous().view(-1)
        out = torch.empty_like(m)
        ops.local_reparam_fwd(m, v, out, m.numel(), eps=e, seed=seed, stream_id=stream_id)
        ctx.save_for_backward(v, e)
        ctx.meta = (seed, stream_id, ops, mean.shape)
        return out.view(mean.shape)

    @staticmethod
    @once_differentiable          # the kernels produce plain tensors: no double backward
    def backward(ctx, grad_out):
        v, e = ctx.saved_tensors
        seed, stream_id, ops, shape = ctx.meta
        g = grad_out.contiguous().view(-1)
        gvar = torch.empty_like(g)
        ops.local_reparam_bwd(g, v, gvar, g.numel(), eps=e, seed=seed, stream_id=stream_id)
        return grad_out, gvar.view(shape), None, None, None, None


class _VarOperand(torch.autograd.Function):
    """An operand of the variance product, one pass (bde_var_operand_fwd/bwd) -- mode 0: clamp(x^2, 1e-4);
    1: clamp(softplus(rho)^2, 1e-4); 2: softplus(rho)^2 (bbb_layers.py:66-67,71,150-153)."""

    @staticmethod
    def forward(ctx, v, mode, ops):
        vc = v.detach().contiguous()
        out = torch.empty_like(vc)
        ops.var_operand_fwd(vc, mode, out)
        ctx.save_for_backward(vc)
        ctx.meta = (mode, ops)
        return out.view(v.shape)

    @staticmethod
    @once_differentiable          # the kernels produce plain tensors: no double backward
    def backward(ctx, grad_out):
        (vc,) = ctx.saved_tensors
        mode, ops = ctx.meta
        gv = torch.empty_like(vc)
        ops.var_operand_bwd(grad_out.contiguous(), vc, mode, gv)
        return gv.view(grad_out.shape), None, None


class _LrtLinear(torch.autograd.Function):
    """The whole forward of a mean-field linear layer in local-reparameterisation form (bbb_layers.py:61-80) as ONE
    fused op (bde_lrt_linear_fwd: the weights are streamed once, sigma^2 and x^2 formed on the fly, both products on
    the MFMA); the backward is bde_lrt_linear_bwd (the autograd graph of those lines in three launches, from the saved
    variance and noise)."""

    @staticmethod
    def forward(ctx, x, w_mu, w_rho, b_mu, b_rho, clamp_bias, eps, seed, stream_id, ops, w_s2=None, w_ds2=None):
        x2d = x.reshape(-1, x.shape[-1])
        if x2d.stride(-1) != 1:
            x2d = x2d.contiguous()
        b, o = x2d.shape[0], w_mu.shape[0]
        out = torch.empty((b, o), dtype=torch.float32, device=x.device)
        var = torch.empty_like(out)
        e = None if eps is None else eps.reshape(b, o).contiguous()
        ops.lrt_linear_fwd(x2d.detach(), w_mu.detach().contiguous(), w_rho.detach().contiguous(),
                           None if b_mu is None else b_mu.detach(), None if b_rho is None else b_rho.detach(),
                           clamp_bias, out, var, eps=e, seed=seed, stream_id=stream_id, w_s2=w_s2)
        ctx.save_for_backward(x2d, w_mu, w_rho, b_rho, var, e, w_s2, w_ds2)
        ctx.meta = (clamp_bias, seed, stream_id, ops, x.shape)
        return out.view(x.shape[:-1] + (o,))

    @staticmethod
    @once_differentiable          # the kernels produce plain tensors: no double backward
    def backward(ctx, grad_out):
        x, w_mu, w_rho, b_rho, var, eps, w_s2, w_ds2 = ctx.saved_tensors
        clamp_bias, seed, stream_id, ops, x_shape = ctx.meta
        g = grad_out.reshape(var.shape).contiguous()
        g_x = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device) if ctx.needs_input_grad[0] else None
        g_wmu, g_wrho = torch.empty_like(w_mu), torch.empty_like(w_rho)
        g_bmu = g_brho = None
        if b_rho is not None:
            g_bmu, g_brho = torch.empty_like(b_rho), torch.empty_like(b_rho)
        # eps None: the kernel regenerates the forward's in-kernel noise (same element indexing)
        ops.lrt_linear_bwd(x, w_mu.detach().contiguous(), w_rho.detach().contiguous(), None if b_rho is None else b_rho.detach(),
                           clamp_bias, g, var, g_x, g_wmu, g_wrho, g_bmu, g_brho, eps=eps, seed=seed, stream_id=stream_id,
                           w_s2=w_s2, w_ds2=w_ds2)
        return (None if g_x is None else g_x.view(x_shape)), g_wmu, g_wrho, g_bmu, g_brho, None, None, None, None, None, \
            None, None


class _ConvWeights:
    """The prepared weight buffer of a fused BBBConv2d (bde_conv_lrt_prep: sigma^2, its rho-derivative and both weight
    matrices in the kernels' staging order), computed ONCE per version of (mean, rho): the weights change at
    ``base_optimizer.step()``, a BBB step runs ``mc_samples`` forward and backward passes per version (bbb.py:63-67).
    Same key as ``_SigmaCache`` (addresses, autograd version counters, the process-wide epoch BBBOptimizer advances); a
    refresh fills a NEW buffer, so a forward whose backward has not run yet keeps the buffer of its own version."""

    def __init__(self):
        self.key = None
        self.buf = None

    def drop(self) -> None:
        self.key = None

    def get(self, mean: torch.Tensor, rho: torch.Tensor, b_rho, ops, stride=(1, 1), padding=(0, 0)):
        """``stride`` / ``padding``: the layer's; a strided layer's buffer also carries the per-phase matrices of its input
        gradient (bde_conv_lrt_prep_strided)."""
        key = (mean.data_ptr(), mean._version, rho.data_ptr(), rho._version, tuple(rho.shape), _SigmaCache.epoch,
               None if b_rho is None else (b_rho.data_ptr(), b_rho._version), tuple(stride), tuple(padding))
        if key != self.key:
            buf = ops.conv_lrt_wbuf(rho.shape, rho.device)
            ops.conv_lrt_prep(mean.detach().contiguous(), rho.detach().contiguous(), buf,
                              None if b_rho is None else b_rho.detach().contiguous(), stride=stride, padding=padding)
            self.key, self.buf = key, buf
        return self.buf


def _conv_profitable(x_shape, w_shape, stride, padding, ops, needs_grad) -> bool:
    """fused_conv="auto": fused only where conv_profit.json records a device measurement of THIS kernel version that beats
    the stock sequence for the pass at hand (forward, or forward + backward)."""
    from . import conv_profit
    abi = getattr(ops, "_abi_version", None)
    if abi is None:
        abi = int(ops.lib.bde_version()) if hasattr(getattr(ops, "lib", None), "bde_version") else -1
        try:
            ops._abi_version = abi
        except AttributeError:
            pass
    if hasattr(ops, "conv_lrt_set_tiling"):
        conv_profit.apply_tilings(ops, tuple(x_shape), tuple(w_shape), stride, padding, abi)    # the autotuned tilings, once
    return conv_profit.profitable(tuple(x_shape), tuple(w_shape), stride, padding, needs_grad, abi)


class _ConvLrt(torch.autograd.Function):
    """The whole forward of a mean-field convolution layer in local-reparameterisation form (bbb_layers.py:146-154) as
    ONE fused op (bde_conv_lrt_fwd: both convolutions as one dual-accumulator implicit GEMM over the same staged input
    windows, the sampling epilogue fused); the backward is bde_conv_lrt_gvar_bias (g_var + both bias gradients, one pass) +
    bde_conv_lrt_bwd_data + bde_conv_lrt_bwd_weight instead of autograd's four convolutions and ~20 element-wise launches."""

    @staticmethod
    def forward(ctx, x, w_mu, w_rho, b_mu, b_rho, stride, padding, eps, seed, stream_id, ops, wbuf, phases=False, want_var=True):
        xc = x.detach().contiguous()
        n, o = xc.shape[0], w_mu.shape[0]
        ho = (xc.shape[2] + 2 * padding[0] - w_mu.shape[2]) // stride[0] + 1
        wo = (xc.shape[3] + 2 * padding[1] - w_mu.shape[3]) // stride[1] + 1
        out = torch.empty((n, o, ho, wo), dtype=torch.float32, device=x.device)
        # the total variance is what the backward needs (sqrt(var)): a forward nobody will differentiate does not write it
        var = torch.empty_like(out) if want_var else None
        e = None if eps is None else eps.reshape(out.shape).contiguous()
        # the bias variance softplus(b_rho)^2 (not clamped, line 147) was evaluated by the preparation pass
        ops.conv_lrt_fwd(xc, wbuf, tuple(w_mu.shape), None if b_mu is None else b_mu.detach().contiguous(), b_rho is not None,
                         stride, padding, out, var, eps=e, seed=seed, stream_id=stream_id)
        ctx.save_for_backward(xc, w_mu, w_rho, b_rho, var, e, wbuf)
        ctx.meta = (stride, padding, seed, stream_id, ops, bool(phases))     # phases: wbuf was prepared with this stride / padding
        return out

    @staticmethod
    @once_differentiable          # the kernels produce plain tensors: no double backward
    def backward(ctx, grad_out):
        x, w_mu, w_rho, b_rho, var, eps, wbuf = ctx.saved_tensors
        stride, padding, seed, stream_id, ops, phases = ctx.meta
        g = grad_out.contiguous()
        gvar = torch.empty_like(g)
        g_bmu = g_brho = None
        if b_rho is not None:
            g_bmu, g_brho = torch.empty_like(b_rho), torch.empty_like(b_rho)
        # ONE pass: g_var and (with a bias) both bias gradients -- the channel sums of g and g_var ride on it, the rho chain rule
        # in its finish.  eps None: the kernel regenerates the forward's in-kernel noise (same element numbering)
        ops.conv_lrt_gvar_bias(g, var, gvar, eps=eps, seed=seed, stream_id=stream_id,
                               b_rho=None if b_rho is None else b_rho.detach().contiguous(), g_bmu=g_bmu, g_brho=g_brho)
        g_x = None
        if ctx.needs_input_grad[0]:
            g_x = torch.empty_like(x)
            ops.conv_lrt_bwd_data(g, gvar, wbuf, tuple(w_mu.shape), x, g_x, stride, padding, phases=phases)
        wr = w_rho.detach().contiguous()
        g_wmu, g_wrho = torch.empty_like(wr), torch.empty_like(wr)
        ops.conv_lrt_bwd_weight(x, g, gvar, wr, g_wmu, g_wrho, stride, padding)
        return g_x, g_wmu, g_wrho, g_bmu, g_brho, None, None, None, None, None, None, None, None, None


class _LocalReparamLayer(nn.Module):
    """Shared machinery: Gaussian weight/bias, noise policy, lazy KL."""

    def _init_common(self, weight_shape, bias_shape, weight_prior, bias_prior, kwargs):
        self.is_bayesian = True
        self.sampling = kwargs.get("sampling", "activations")
        self.freeze_on_eval = kwargs.get("freeze_on_eval", True)
        self.kl_on_eval = kwargs.get("kl_on_eval", False)
        self.use_bias = kwargs.get("bias", True)
        self.fused_epilogue = kwargs.get("fused_epilogue", True)     # one HIP pass for mean + sqrt(var) * eps
        self.fused_linear = kwargs.get("fused_linear", True)         # BBBLinear: the whole forward as one fused op
        self.sigma_cache = kwargs.get("sigma_cache", True)           # wide BBBLinear: sigma^2 once per weight version
        # BBBLinear: the fused op takes at most 128 rows per launch (_LRT_TILE_ROWS).  A batch of up to this many rows is run as
        # ceil(rows / 128) launches of the SAME device-verified kernels; above it the two stock GEMMs + fused element-wise
        # passes.  Default 128 = one launch, the only form with a device measurement against the stock sequence (round 3);
        # bench.py's bbb_linear_*_tiled entries measure 256 / 512 / 1024 rows so that a device run can raise it.
        self.fused_linear_max_rows = int(kwargs.get("fused_linear_max_rows", _LRT_TILE_ROWS))
        # BBBConv2d: both convolutions + epilogue as one fused op.  "auto" (default): only at geometries where a DEVICE
        # measurement shows the fused kernels ahead of the stock sequence (conv_profit.py); True: wherever the kernels
        # have a tiling; False: never (stock convolutions + fused element-wise passes)
        self.fused_conv = kwargs.get("fused_conv", "auto")
        self._sigma_cache = _SigmaCache()
        self._conv_weights = _ConvWeights()
        self.weight_prior, self.bias_prior = weight_prior, bias_prior
        gp_kwargs = {k: kwargs[k] for k in ("rng", "seed", "_ops") if k in kwargs}
        self.weight = GaussianParameter(weight_shape, **gp_kwargs)
        if self.use_bias:
            self.bias = GaussianParameter(bias_shape, **gp_kwargs)
        self.reset_parameters()

    def reset_parameters(self):
        self.weight.blundell_init()
        if self.use_bias:
            self.bias.blundell_init()

    def invalidate_sigma_cache(self) -> None:
        """Drop this layer's cached sigma^2 / prepared convolution weights (after an edit of ``rho.data`` that the
        version counter cannot see).  A method, not a closure: layers stay picklable and a deep copy drops ITS caches."""
        self._sigma_cache.drop()
        self._conv_weights.drop()

    def __getstate__(self):
        # the caches hold device buffers keyed on tensor addresses of THIS process: a copy / an unpickled layer starts cold
        state = dict(self.__dict__)
        state["_sigma_cache"] = _SigmaCache()
        state["_conv_weights"] = _ConvWeights()
        return state

    @property
    def kl(self):
        if not (self.training or self.kl_on_eval):
            return 0
        total = self.weight_prior.kl_divergence(self.weight.mean, self.weight.std)
        if self.use_bias:
            total = total + self.bias_prior.kl_divergence(self.bias.mean, self.bias.std)
        return total

    def _var_operands(self, input: torch.Tensor, clamp_bias: bool):
        """clamp(x^2), clamp(sigma_W^2) and the bias variance of the variance product (bbb_layers.py:66-67,71,150-153):
        one fused pass each on fp32 device tensors, the reference's op sequence otherwise."""
        w, b = self.weight, (self.bias if self.use_bias else None)
        fusable = self.fused_epilogue and input.dtype == torch.float32 and input.is_cuda == w.mean.is_cuda
        ops = w._get_ops() if fusable else None
        fuse_min = _fuse_min(ops) if fusable else 0

        def operand(t, mode):
            if fusable and t.numel() >= fuse_min:
                return _var_operand(t, mode, ops)
            if mode == 0:
                return (t ** 2).clamp(min=_CLAMP)
            s2 = F.softplus(t) ** 2
            return s2.clamp(min=_CLAMP) if mode == 1 else s2
        vb = None if b is None else operand(b.rho, 1 if clamp_bias else 2)
        return operand(input, 0), operand(w.rho, 1), vb

    def _noise(self, mean: torch.Tensor) -> torch.Tensor:
        if not self.training and self.freeze_on_eval:
            # one draw for the whole batch, so an eval pass uses ONE set of weights per call
            eps = torch.empty(mean.shape[1:], device=mean.device, dtype=mean.dtype).normal_(0, 1)
            return eps.unsqueeze(0).expand(mean.shape)
        return normal_like(mean)

    def _sample_activations(self, mean: torch.Tensor, var: torch.Tensor) -> torch.Tensor:
        frozen = not self.training and self.freeze_on_eval
        if frozen or not self.fused_epilogue or mean.dtype != torch.float32:
            return mean + torch.sqrt(var) * self._noise(mean)       # stock PyTorch (eval: broadcast noise)
        gp = self.weight
        philox = gp.rng == "philox" and gp.noise_source is None
        if mean.numel() < _fuse_min(gp._get_ops()):
            # small activations, Python nodes only: ATen nodes are cheaper; the noise stays the layer's Philox stream
            if philox:
                eps = torch.empty_like(mean, memory_format=torch.contiguous_format)
                gp._get_ops().philox_normal(gp.seed, next(_philox_stream), eps_d=eps.view(-1), d=eps.numel())
            else:
                eps = normal_like(mean)
            return mean + torch.sqrt(var) * eps
        eps = None if philox else normal_like(mean)
        return _local_reparam(mean, var, eps, gp.seed, next(_philox_stream), gp._get_ops())


class BBBLinear(_LocalReparamLayer):
    def __init__(self, in_features: int, out_features: int, weight_prior, bias_prior, **kwargs):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.mc_sample = kwargs.get("mc_sample", 1)
        self.rho_init = kwargs.get("rho_init", -3)
        self._init_common((out_features, in_features), (out_features,), weight_prior, bias_prior, kwargs)

    def forward(self, input: torch.Tensor):
        if self.sampling == "activations":
            w, b = self.weight, (self.bias if self.use_bias else None)
            frozen = not self.training and self.freeze_on_eval       # eval: ONE noise draw shared by the batch
            rows = input.numel() // max(1, input.shape[-1])
            tiled = rows > _LRT_TILE_ROWS
            if self.fused_linear and not frozen and input.dtype == torch.float32 and input.is_cuda == w.mean.is_cuda \
                    and 1 <= rows <= max(_LRT_TILE_ROWS, getattr(self, "fused_linear_max_rows", _LRT_TILE_ROWS)) \
                    and w._get_ops().lrt_linear_supported(min(rows, _LRT_TILE_ROWS), self.in_features, self.out_features):
                eps = None
                if not (w.rng == "philox" and w.noise_source is None):
                    eps = normal_like(input.new_empty(input.shape[:-1] + (self.out_features,)))
                ops = w._get_ops()
                s2 = ds2 = None
                if self.sigma_cache and ops.lrt_sigma_cache_wanted(self.in_features, self.out_features):
                    s2, ds2 = self._sigma_cache.get(w.rho, ops)
                if tiled:
                    out = _lrt_linear_tiled(input, w.mean, w.rho, b.mean if b is not None else None,
                                            b.rho if b is not None else None, True, eps, w.seed, lambda: next(_philox_stream),
                                            ops, s2, ds2)
                else:
                    out = _lrt_linear(input, w.mean, w.rho, b.mean if b is not None else None,
                                      b.rho if b is not None else None, True, eps, w.seed, next(_philox_stream), ops, s2, ds2)
                return out / self.mc_sample
            mean = F.linear(input, w.mean, b.mean if b is not None else None)
            var = F.linear(*self._var_operands(input, clamp_bias=True))
            return self._sample_activations(mean, var) / self.mc_sample
        if self.sampling == "parameters":
            out = None
            for _ in range(self.mc_sample):
                y = F.linear(input, self.weight.sample(), self.bias.sample() if self.use_bias else None)
                out = y if out is None else out + y
            return out / self.mc_sample
        raise ValueError("Invalid value of sampling")

    def means(self):
        return torch.cat([self.weight.mean.flatten(), self.bias.mean.flatten()])

    def sigmas(self):
        return torch.cat([self.weight.std.flatten(), self.bias.std.flatten()])


class BBBConv2d(_LocalReparamLayer):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, weight_prior, bias_prior, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride = kwargs.get("stride", 1)
        self.padding = kwargs.get("padding", 0)
        self._init_common((out_channels, in_channels, kernel_size, kernel_size), (out_channels,), weight_prior,
                          bias_prior, kwargs)

    def forward(self, input: torch.Tensor):
        if self.sampling == "parameters":
            raise NotImplementedError()
        if self.sampling != "activations":
            raise ValueError("Invalid value of sampling")
        w, b = self.weight, (self.bias if self.use_bias else None)
        frozen = not self.training and self.freeze_on_eval           # eval: ONE noise draw shared by the batch (stock path)
        if self.fused_conv and not frozen and input.dim() == 4 and input.dtype == torch.float32 \
                and not isinstance(self.padding, str) and not isinstance(self.stride, str) \
                and input.is_cuda == w.mean.is_cuda and w.mean.is_contiguous() and w.rho.is_contiguous() \
                and hasattr(w._get_ops(), "conv_lrt_fwd"):
            ops = w._get_ops()
            stride, padding = _pair(self.stride), _pair(self.padding)
            # (_conv_profitable also pins the tilings tools/conv_autotune.py recorded for this layer, forced path included)
            if ops.conv_lrt_supported(input.shape, w.mean.shape, stride, padding) and (
                    _conv_profitable(input.shape, w.mean.shape, stride, padding, ops,
                                     torch.is_grad_enabled() and (input.requires_grad or w.mean.requires_grad
                                                                  or w.rho.requires_grad))
                    or self.fused_conv is True):
                eps = None
                if not (w.rng == "philox" and w.noise_source is None):
                    ho = (input.shape[2] + 2 * padding[0] - self.kernel_size) // stride[0] + 1
                    wo = (input.shape[3] + 2 * padding[1] - self.kernel_size) // stride[1] + 1
                    eps = normal_like(input.new_empty((input.shape[0], self.out_channels, ho, wo)))
                wbuf = self._conv_weights.get(w.mean, w.rho, b.rho if b is not None else None, ops, stride, padding)
                native = _native_nodes(ops)
                want_var = torch.is_grad_enabled() and (input.requires_grad or any(
                    t is not None and t.requires_grad for t in (w.mean, w.rho, b.mean if b is not None else None,
                                                                b.rho if b is not None else None)))
                if native is not None and hasattr(native, "conv_lrt"):
                    return native.conv_lrt(input, w.mean, w.rho, b.mean if b is not None else None,
                                           b.rho if b is not None else None, stride[0], stride[1], padding[0], padding[1], eps,
                                           w.seed, next(_philox_stream), wbuf, True, want_var)
                return _ConvLrt.apply(input, w.mean, w.rho, b.mean if b is not None else None,
                                      b.rho if b is not None else None, stride, padding, eps, w.seed, next(_philox_stream), ops,
                                      wbuf, True, want_var)
        mean = F.conv2d(input, w.mean, b.mean if b is not None else None, stride=self.stride, padding=self.padding)
        x2, s2, vb = self._var_operands(input, clamp_bias=False)     # the conv layer does not clamp its bias variance
        var = F.conv2d(x2, s2, vb, stride=self.stride, padding=self.padding)
        return self._sample_activations(mean, var)


def make_module_bbb(module: nn.Module, prior, **kwargs) -> int:
    """Replace every nn.Linear / nn.Conv2d of ``module`` by its mean-field counterpart, keeping the
    current weights as the means (cf. the reference's ``make_module_bbb``); returns the number of
    layers replaced."""
    replaced = 0
    for name, child in list(module.named_children()):
        new = None
        if isinstance(child, nn.Linear):
            new = BBBLinear(child.in_features, child.out_features, prior, prior, bias=child.bias is not None, **kwargs)
        elif isinstance(child, nn.Conv2d) and child.groups == 1 and child.dilation == (1, 1) \
                and child.kernel_size[0] == child.kernel_size[1] and isinstance(child.padding, tuple):
            new = BBBConv2d(child.in_channels, child.out_channels, child.kernel_size[0], prior, prior,
                            stride=child.stride, padding=child.padding, bias=child.bias is not None, **kwargs)
        if new is not None:
            new = new.to(child.weight.device)
            with torch.no_grad():
                new.weight.mean.copy_(child.weight)
                if child.bias is not None:
                    new.bias.mean.copy_(child.bias)
            setattr(module, name, new)
            replaced += 1
        else:
            replaced += make_module_bbb(child, prior, **kwargs)
    return replaced
