"""Tensor-level wrappers over the C ABI (include/bde_hip.h).

``HipOps`` is the only kernel backend the product has.  Each method checks its
tensors (CUDA, fp32, contiguous last dim), passes raw device pointers and the
CURRENT torch stream to libbde_hip.so and raises on a non-zero return code.
The optimizer shells call kernels only through an object with this interface;
tests may inject a checker object with the same methods (see tests/), the
product never does.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib


class BdeKernelError(RuntimeError):
    pass


def _ptr(t: Optional[torch.Tensor], name: str = "tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise BdeKernelError(f"{name}: expected a CUDA (HIP) tensor, got device {t.device}; "
                             "beyond_deep_ensembles_amd has no CPU path")
    if t.dtype != torch.float32:
        raise BdeKernelError(f"{name}: expected float32, got {t.dtype}")
    if t.dim() > 0 and t.stride(-1) != 1:
        raise BdeKernelError(f"{name}: last dimension must be contiguous")
    return t.data_ptr()


def _ptr64(t: torch.Tensor, name: str):
    """Device pointer of a float64 tensor (the fp64 Gram blocks of the dimension-sharded exchange)."""
    if not t.is_cuda or t.dtype != torch.float64:
        raise BdeKernelError(f"{name}: expected a CUDA (HIP) float64 tensor, got {t.dtype} on {t.device}")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw hipStream_t of torch's current stream on the current device (the binding torch.cuda.current_stream()
    wraps; building the Stream object costs ~10 us per call, a third of a small op's host time)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _on_device_of(method):
    """Run a HipOps method with the device of its first tensor argument current, so that the kernel is
    enqueued on THAT device's current torch stream (multi-GPU processes)."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        dev = None
        for a in args:
            if torch.is_tensor(a):
                dev = a.device
                break
        if dev is None or dev.type != "cuda" or dev.index == torch.cuda.current_device():
            return method(self, *args, **kwargs)
        with torch.cuda.device(dev):
            return method(self, *args, **kwargs)
    return wrapper


def _check(rc: int, what: str):
    if rc != 0:
        raise BdeKernelError(f"{what} failed with code {rc}" + (" (invalid argument)" if rc == -1 else " (hipError)"))


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.dim() == 2 else t.shape[-1]


def pad4(n: int, mult: int = 64) -> int:
    """Leading dimension used for flat rows: a multiple of 64 floats (256 B)."""
    return (n + mult - 1) // mult * mult


class SegTable:
    """Where the gradients of the M particles live, per parameter tensor ("segment"), for the ``*_seg`` entry points
    (include/bde_hip.h): ``ptrs`` (device int64 ``[n_seg * M]``, refreshed every step from ``host``), ``chunks``
    (device, static: pieces of <= 256 float4 columns of one segment).  ``offsets`` (multiples of 4) / ``numels`` describe
    the segments on the host."""

    def __init__(self, offsets, numels, m, device):
        import numpy as np
        if any(o % 4 for o in offsets):
            raise BdeKernelError("SegTable: every segment must start on a float4 boundary of the row (FlatLayout(align=4))")
        self.offsets, self.numels, self.m = list(offsets), list(numels), int(m)
        self.n_seg = len(self.offsets)
        rows = []
        for s, (col0, n) in enumerate(zip(self.offsets, self.numels)):
            for start in range(0, n, 1024):
                nflt = min(1024, n - start)
                rows.append(((col0 + start) // 4, start // 4, s + (nflt << 32), 0))
        self.n_chunks = len(rows)
        # flat columns [start, end) each piece covers, for callers that pack a COLUMN RANGE of the rows (piece_range)
        self.piece_start = np.asarray([4 * r[0] for r in rows], dtype=np.int64)
        self.piece_end = self.piece_start + np.asarray([r[2] >> 32 for r in rows], dtype=np.int64)
        self.chunks = torch.from_numpy(np.asarray(rows, dtype=np.int64).reshape(-1, 4)).to(device)
        self.ptrs = torch.zeros(max(1, self.n_seg * self.m), dtype=torch.int64, device=device)
        on_gpu = torch.device(device).type == "cuda"
        # host staging, rotated so that a table is never overwritten while its upload may still be in flight
        self.host = [torch.zeros(max(1, self.n_seg * self.m), dtype=torch.int64) for _ in range(3)]
        if on_gpu:
            self.host = [h.pin_memory() for h in self.host]
        self._events = [None] * len(self.host)
        self._event_pool = [None] * len(self.host)
        self._slot = 0

    def piece_range(self, c0: int, c1: int):
        """Indices [q0, q1) of the pieces that intersect flat columns [c0, c1) (pieces are sorted by column)."""
        import numpy as np
        q0 = int(np.searchsorted(self.piece_end, c0, side="right"))
        q1 = int(np.searchsorted(self.piece_start, c1, side="left"))
        return q0, max(q0, q1)

    def upload_again(self) -> None:
        """Re-send the host table being filled (more entries are valid now) without closing the step's staging slot."""
        self.ptrs.copy_(self.host[self._slot], non_blocking=True)

    def staging(self) -> torch.Tensor:
        """The host table to fill for the coming step."""
        ev = self._events[self._slot]
        if ev is not None:
            ev.synchronize()
            self._events[self._slot] = None
        return self.host[self._slot]

    def upload(self) -> None:
        """Host table -> device (stream-ordered, asynchronous from pinned memory)."""
        self.ptrs.copy_(self.host[self._slot], non_blocking=True)
        self.uploaded()

    def uploaded(self) -> None:
        """The current staging slot's copy has been enqueued (by upload(), or by a replayed graph that contains it): mark
        the slot busy until that point of the stream and move on to the next one."""
        if self.ptrs.is_cuda:
            ev = self._event_pool[self._slot]                 # one event per slot, re-recorded (creating one costs ~2 us a step)
            if ev is None:
                ev = self._event_pool[self._slot] = torch.cuda.Event()
            ev.record()
            self._events[self._slot] = ev
        self._slot = (self._slot + 1) % len(self.host)


class HipOps:
    """Kernel backend over libbde_hip.so."""

    name = "hip"

    _loaded_devices = set()          # devices whose code objects bde_init() has loaded (process-wide)

    def __init__(self):
        self.lib = _lib.load()
        if torch.cuda.is_available():
            self.load_code_objects(torch.cuda.current_device())

    def load_code_objects(self, device) -> None:
        """``bde_init()`` on ``device``: every code object of the library is uploaded NOW, from this thread, instead of
        at the first launch of one of its kernels (HIP's default).  Done once per device and process, when the first
        ``HipOps`` is built and again by the shells for the device their parameters live on -- i.e. before a process
        group's communication threads exist and before the first collective (profiles/r03_first_launch_*.txt)."""
        index = torch.device(device).index if not isinstance(device, int) else device
        if index is None:
            index = torch.cuda.current_device()
        if index in HipOps._loaded_devices:
            return
        with torch.cuda.device(index):
            _check(self.lib.bde_init(), "bde_init")
            failed = int(self.lib.bde_init_optional_failures())
        if failed:
            import warnings
            names = [n for bit, n in enumerate(("svgd_small.hip", "conv_lrt.hip", "conv_lrt_bwd.hip")) if failed >> bit & 1]
            warnings.warn(f"libbde_hip.so: the code object(s) of {', '.join(names)} did not load on device {index}; their kernels "
                          "are opt-in (device_verified.py / conv_profit.py) and will fail when launched -- every default path is unaffected")
        HipOps._loaded_devices.add(index)

    # ------------------------------------------------------------ SVGD --

    def svgd_ws(self, m: int, device) -> torch.Tensor:
        n = self.lib.bde_svgd_ws_bytes(m)
        if n == 0:
            raise BdeKernelError(f"SVGD supports 1 <= particle_count <= 64, got {m}")
        return torch.zeros(n // 4, dtype=torch.float32, device=device)    # the header must start out zero

    def svgd_kstat(self, m: int, device) -> torch.Tensor:
        return torch.zeros(self.lib.bde_svgd_kstat_floats(m), dtype=torch.float32, device=device)

    @_on_device_of
    def svgd_gram(self, P, d, ws):
        m = P.shape[0]
        _check(self.lib.bde_svgd_gram(_ptr(P, "P"), m, d, _ld(P), _ptr(ws), _stream()), "bde_svgd_gram")

    def svgd_set_gram_keep_bytes(self, nbytes: int) -> None:
        """Tuning hook (process-wide): how much of the Gram pass's tail stays cacheable for the combine pass."""
        _check(self.lib.bde_svgd_set_gram_keep_bytes(int(nbytes)), "bde_svgd_set_gram_keep_bytes")

    @_on_device_of
    def svgd_kstats(self, ws, m, l2_reg, kernel_grad_scale, dataset_size, sign, kstat, h_override=0.0, mode=0):
        _check(self.lib.bde_svgd_kstats(_ptr(ws), m, l2_reg, kernel_grad_scale, dataset_size, sign, h_override, mode,
                                        _ptr(kstat), _stream()), "bde_svgd_kstats")

    GMAT_DOUBLES = 257

    @_on_device_of
    def svgd_gram_finish(self, ws, m, gmat_out):
        """Partials of ``ws`` -> fp64 Gram block ``gmat_out [257]`` (dimension-sharded exchange)."""
        if gmat_out.numel() < self.GMAT_DOUBLES:
            raise BdeKernelError("gmat_out: expected a CUDA float64 tensor with >= 257 elements")
        _check(self.lib.bde_svgd_gram_finish(_ptr(ws), m, _ptr64(gmat_out, "gmat_out"), _stream()), "bde_svgd_gram_finish")

    @_on_device_of
    def svgd_kstats_gmat(self, gmats, m, l2_reg, kernel_grad_scale, dataset_size, sign, kstat, h_override=0.0, mode=0):
        """Statistics from the ranks' Gram blocks ``gmats [n, >= 257]`` (float64), summed in row order."""
        if gmats.dim() != 2 or gmats.stride(1) != 1:
            raise BdeKernelError("gmats: expected a CUDA float64 [n, >= 257] tensor")
        _check(self.lib.bde_svgd_kstats_gmat(_ptr64(gmats, "gmats"), gmats.shape[0], gmats.stride(0), m, l2_reg,
                                             kernel_grad_scale, dataset_size, sign, h_override, mode, _ptr(kstat),
                                             _stream()), "bde_svgd_kstats_gmat")

    @_on_device_of
    def svgd_combine(self, P, G, out, d, kstat):
        """out = CG @ G + CP @ P over the first d columns; P / out may be column-offset views of the flat buffers
        (same row stride), G may have its own row stride (a staging buffer of the multi-GPU exchange)."""
        m = P.shape[0]
        if _ld(out) != _ld(P):
            raise BdeKernelError("P and out must share one leading dimension")
        _check(self.lib.bde_svgd_combine(_ptr(P, "P"), _ptr(G, "G"), _ptr(out, "out"), m, d, _ld(P),
                                         _ld(G) if G is not None else 0, _ptr(kstat), _stream()), "bde_svgd_combine")

    @_on_device_of
    def svgd_step(self, P, G, out, d, l2_reg, kernel_grad_scale, dataset_size, sign, ws, kstat):
        """out = sign * phi (svgd.py:86-89) by the three streaming launches, at every size; out may alias G.  (The
        small-model kernel is svgd_step_small: an explicit choice, see device_verified.py.)"""
        m = P.shape[0]
        if _ld(G) != _ld(P) or _ld(out) != _ld(P):
            raise BdeKernelError("P, G, out must share one leading dimension")
        _check(self.lib.bde_svgd_step(_ptr(P, "P"), _ptr(G, "G"), _ptr(out, "out"), m, d, _ld(P), l2_reg,
                                      kernel_grad_scale, dataset_size, sign, _ptr(ws), _ptr(kstat), _stream()),
               "bde_svgd_step")

    def svgd_small_supported(self, m: int, d: int) -> bool:
        return bool(self.lib.bde_svgd_small_supported(m, d))

    @_on_device_of
    def svgd_step_small(self, P, G, out, d, l2_reg, kernel_grad_scale, dataset_size, sign, ws, kstat, h_override=0.0,
                        mode=0):
        """The whole update for small models (two launches of one kernel); mode 1 = rbf's grad_kernel (G may be None)."""
        m = P.shape[0]
        if (G is not None and _ld(G) != _ld(P)) or _ld(out) != _ld(P):
            raise BdeKernelError("P, G, out must share one leading dimension")
        _check(self.lib.bde_svgd_step_small(_ptr(P, "P"), _ptr(G, "G"), _ptr(out, "out"), m, d, _ld(P), l2_reg,
                                            kernel_grad_scale, dataset_size, sign, h_override, mode, _ptr(ws),
                                            _ptr(kstat), _stream()),
               "bde_svgd_step_small")

    @_on_device_of
    def svgd_step_small_sgd(self, P, G, buf, d, l2_reg, kernel_grad_scale, dataset_size, ws, kstat, lr, momentum,
                            dampening, weight_decay, nesterov, first):
        """Whole step (statistics, -phi, M shared-state SGD applications, particles updated in place)."""
        if _ld(G) != _ld(P):
            raise BdeKernelError("P and G must share one leading dimension")
        _check(self.lib.bde_svgd_step_small_sgd(_ptr(P, "P"), _ptr(G, "G"), _ptr(buf), P.shape[0], d, _ld(P), l2_reg,
                                                kernel_grad_scale, dataset_size, lr, momentum, dampening, weight_decay,
                                                int(nesterov), int(first), _ptr(ws), _ptr(kstat), _stream()),
               "bde_svgd_step_small_sgd")

    @_on_device_of
    def svgd_step_small_adam(self, P, G, exp_avg, exp_avg_sq, d, l2_reg, kernel_grad_scale, dataset_size, ws, kstat, lr,
                             beta1, beta2, eps, weight_decay, step0):
        if _ld(G) != _ld(P):
            raise BdeKernelError("P and G must share one leading dimension")
        _check(self.lib.bde_svgd_step_small_adam(_ptr(P, "P"), _ptr(G, "G"), _ptr(exp_avg), _ptr(exp_avg_sq), P.shape[0],
                                                 d, _ld(P), l2_reg, kernel_grad_scale, dataset_size, lr, beta1, beta2, eps,
                                                 weight_decay, int(step0), _ptr(ws), _ptr(kstat), _stream()),
               "bde_svgd_step_small_adam")

    @_on_device_of
    def svgd_apply_sgd(self, P, grad, buf, d, lr, momentum, dampening, weight_decay, nesterov, first):
        _check(self.lib.bde_svgd_apply_sgd(_ptr(P), _ptr(grad), _ptr(buf), P.shape[0], d, _ld(P), lr, momentum,
                                           dampening, weight_decay, int(nesterov), int(first), _stream()),
               "bde_svgd_apply_sgd")

    @_on_device_of
    def svgd_apply_adam(self, P, grad, exp_avg, exp_avg_sq, d, lr, beta1, beta2, eps, weight_decay, step0):
        _check(self.lib.bde_svgd_apply_adam(_ptr(P), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), P.shape[0], d,
                                            _ld(P), lr, beta1, beta2, eps, weight_decay, int(step0), _stream()),
               "bde_svgd_apply_adam")

    def svgd_fused_gram_supported(self, m: int) -> bool:
        return bool(self.lib.bde_svgd_fused_gram_supported(m))

    @_on_device_of
    def svgd_fused_sgd(self, P, G, buf, d, kstat, lr, momentum, dampening, weight_decay, nesterov, first, ws_next=None):
        """combine + M shared-state SGD applications in one pass; optionally the next step's Gram partials."""
        _check(self.lib.bde_svgd_fused_sgd(_ptr(P, "P"), _ptr(G, "G"), _ptr(buf), P.shape[0], d, _ld(P), _ld(G), _ptr(kstat),
                                           lr, momentum, dampening, weight_decay, int(nesterov), int(first),
                                           _ptr(ws_next), _stream()), "bde_svgd_fused_sgd")

    @_on_device_of
    def svgd_fused_adam(self, P, G, exp_avg, exp_avg_sq, d, kstat, lr, beta1, beta2, eps, weight_decay, step0,
                        ws_next=None):
        _check(self.lib.bde_svgd_fused_adam(_ptr(P, "P"), _ptr(G, "G"), _ptr(exp_avg), _ptr(exp_avg_sq), P.shape[0], d,
                                            _ld(P), _ld(G), _ptr(kstat), lr, beta1, beta2, eps, weight_decay, int(step0),
                                            _ptr(ws_next), _stream()), "bde_svgd_fused_adam")

    # ---- gradients read where autograd left them (no copy into the flat rows) ----
    def seg_table(self, offsets, numels, m, device) -> SegTable:
        return SegTable(offsets, numels, m, device)

    @_on_device_of
    def svgd_combine_seg(self, P, seg: SegTable, out, d, kstat):
        if _ld(out) != _ld(P):
            raise BdeKernelError("P and out must share one leading dimension")
        _check(self.lib.bde_svgd_combine_seg(_ptr(P, "P"), seg.ptrs.data_ptr(), seg.chunks.data_ptr(), seg.n_chunks,
                                             _ptr(out, "out"), P.shape[0], d, _ld(P), _ptr(kstat), _stream()),
               "bde_svgd_combine_seg")

    @_on_device_of
    def svgd_fused_sgd_seg(self, P, seg: SegTable, buf, d, kstat, lr, momentum, dampening, weight_decay, nesterov, first,
                           ws_next=None):
        _check(self.lib.bde_svgd_fused_sgd_seg(_ptr(P, "P"), seg.ptrs.data_ptr(), seg.chunks.data_ptr(), seg.n_chunks,
                                               _ptr(buf), P.shape[0], d, _ld(P), _ptr(kstat), lr, momentum, dampening,
                                               weight_decay, int(nesterov), int(first), _ptr(ws_next), _stream()),
               "bde_svgd_fused_sgd_seg")

    @_on_device_of
    def svgd_fused_adam_seg(self, P, seg: SegTable, exp_avg, exp_avg_sq, d, kstat, lr, beta1, beta2, eps, weight_decay,
                            step0, ws_next=None):
        _check(self.lib.bde_svgd_fused_adam_seg(_ptr(P, "P"), seg.ptrs.data_ptr(), seg.chunks.data_ptr(), seg.n_chunks,
                                                _ptr(exp_avg), _ptr(exp_avg_sq), P.shape[0], d, _ld(P), _ptr(kstat), lr,
                                                beta1, beta2, eps, weight_decay, int(step0), _ptr(ws_next), _stream()),
               "bde_svgd_fused_adam_seg")

    @_on_device_of
    def svgd_gather_seg(self, G, seg: SegTable, row0=0, n_rows=None, pieces=None):
        """Pack the segmented gradients of particles [row0, row0 + n_rows) into the flat rows G [M, ld]: one launch.
        ``pieces = (q0, q1)`` restricts it to those pieces of the table (``SegTable.piece_range``: a column range)."""
        n_rows = seg.m - row0 if n_rows is None else n_rows
        q0, q1 = pieces if pieces is not None else (0, seg.n_chunks)
        if not 0 <= q0 <= q1 <= seg.n_chunks:
            raise BdeKernelError("svgd_gather_seg: piece range outside the table")
        if q0 == q1:
            return
        if row0 < 0 or n_rows < 1 or row0 + n_rows > min(seg.m, G.shape[0]):
            raise BdeKernelError("svgd_gather_seg: rows outside the table / the gradient buffer")
        # M = the table's particle count (its pointer stride), whatever number of rows G holds
        _check(self.lib.bde_svgd_gather_seg(seg.ptrs.data_ptr(), seg.chunks.data_ptr() + 32 * q0, q1 - q0, _ptr(G, "G"),
                                            seg.m, row0, n_rows, _ld(G), _stream()), "bde_svgd_gather_seg")

    def sum_scalars(self, scalars, out) -> None:
        """out[()] = ((s0 + s1) + s2) + ... in fp32, this order, ONE launch (bde_sum_scalars): the returned loss of a
        step (svgd.py:66,72: ``total_loss += loss`` per particle).  ``scalars``: 1..64 fp32 one-element device tensors."""
        self.mean_scalars(scalars, out, 1.0)

    def mean_scalars(self, scalars, out, divisor) -> None:
        """The same sum times fl(1 / ``divisor``) in the same launch (bde_mean_scalars): svgd.py:105 ``total_loss /
        particle_count`` as torch's GPU kernel rounds a division by a Python number."""
        import ctypes
        n = len(scalars)
        if not 1 <= n <= 64:
            raise BdeKernelError("sum_scalars: 1..64 scalars expected")
        if not float(divisor) > 0.0:
            raise BdeKernelError("mean_scalars: the divisor must be positive")
        ptrs = (ctypes.c_void_p * n)()
        for i, t in enumerate(scalars):
            if t.numel() != 1 or t.device != out.device:
                raise BdeKernelError("sum_scalars: one-element tensors on the output's device expected")
            ptrs[i] = _ptr(t, "scalar")
        if out.numel() != 1:
            raise BdeKernelError("sum_scalars: out must hold one element")
        if out.device.type == "cuda" and out.device.index != torch.cuda.current_device():
            with torch.cuda.device(out.device):
                _check(self.lib.bde_mean_scalars(ptrs, n, float(divisor), _ptr(out, "out"), _stream()), "bde_mean_scalars")
            return
        _check(self.lib.bde_mean_scalars(ptrs, n, float(divisor), _ptr(out, "out"), _stream()), "bde_mean_scalars")

    def entry(self, name: str) -> int:
        """Address of C-ABI entry point ``name`` in this backend's library, for the native host helper (csrc/host.cpp
        ``mean_losses`` / ``small_step_*``: argument checks and the calls without Python or ctypes marshalling in between)."""
        import ctypes
        return ctypes.cast(getattr(self.lib, name), ctypes.c_void_p).value

    def mean_scalars_entry(self) -> int:
        return self.entry("bde_mean_scalars")

    # ------------------------------------------------------------ SWAG --
    @_on_device_of
    def swag_update(self, theta, mean, sq, dev_row, n, d):
        _check(self.lib.bde_swag_update(_ptr(theta, "theta"), _ptr(mean), _ptr(sq), _ptr(dev_row), int(n), d, _stream()),
               "bde_swag_update")

    @_on_device_of
    def swag_sample(self, mean, sq, dev, head, out, d, eps_w=None, eps_d=None, seed=0, stream_id=0):
        k = dev.shape[0]
        _check(self.lib.bde_swag_sample(_ptr(mean), _ptr(sq), _ptr(dev), k, _ld(dev), head, _ptr(eps_w), _ptr(eps_d),
                                        seed, stream_id, _ptr(out), d, _stream()), "bde_swag_sample")

    @_on_device_of
    def swag_sample_batched(self, mean, sq, dev, head, out, d, eps_w=None, eps_d=None, seed=0, stream_id0=0):
        """``out``: ``[S, >= d]`` contiguous rows."""
        k, s = dev.shape[0], out.shape[0]
        _check(self.lib.bde_swag_sample_batched(_ptr(mean), _ptr(sq), _ptr(dev), k, _ld(dev), head, _ptr(eps_w),
                                                _ptr(eps_d), _ld(eps_d) if eps_d is not None else 0, seed, stream_id0,
                                                _ptr(out), _ld(out), s, d, _stream()),
               "bde_swag_sample_batched")

    @property
    def swag_philox_rounds(self) -> int:
        """Philox rounds of the SWAG samplers' in-kernel noise (7; every other draw uses the published default, 10)."""
        return int(self.lib.bde_swag_philox_rounds())

    def philox_normal(self, seed, stream_id, eps_w=None, eps_d=None, d=None, rounds=10):
        k = 0 if eps_w is None else eps_w.numel()
        n = 0 if eps_d is None else (d if d is not None else eps_d.numel())
        target = eps_d if eps_d is not None else eps_w
        if target is None:
            return
        if eps_w is not None and eps_d is not None and eps_w.device != eps_d.device:
            raise BdeKernelError("philox_normal: eps_w and eps_d live on different devices")
        pw, pd = _ptr(eps_w, "eps_w"), _ptr(eps_d, "eps_d")
        with torch.cuda.device(target.device):           # the launch goes to the stream of the OUTPUT's device
            _check(self.lib.bde_philox_normal(seed, stream_id, pw, k, pd, n, int(rounds), _stream()), "bde_philox_normal")

    def philox_bits(self, seed, stream_id, n_groups, device, domain=0, idx0=0, rounds=10) -> torch.Tensor:
        """Raw Philox4x32-10 words [n_groups, 4] (int64 holding uint32 values) -- the known-answer hook."""
        out = torch.empty(n_groups * 4, dtype=torch.int32, device=device)
        with torch.cuda.device(out.device):
            _check(self.lib.bde_philox_bits(seed, stream_id, domain, idx0, out.data_ptr(), n_groups, int(rounds), _stream()),
                   "bde_philox_bits")
        return (out.to(torch.int64) & 0xFFFFFFFF).view(n_groups, 4)

    # ----------------------------------------------------------- Gauss --
    def reduce_ws(self, device) -> torch.Tensor:
        return torch.empty(self.lib.bde_reduce_ws_bytes() // 4, dtype=torch.float32, device=device)

    @_on_device_of
    def gauss_draw_fwd(self, mean, rho, out, n, eps=None, seed=0, stream_id=0, eps_out=None):
        _check(self.lib.bde_gauss_draw_fwd(_ptr(mean), _ptr(rho), _ptr(eps), seed, stream_id, _ptr(out), _ptr(eps_out),
                                           n, _stream()), "bde_gauss_draw_fwd")

    @_on_device_of
    def gauss_draw_bwd(self, g, rho, gmean, grho, n, eps=None, seed=0, stream_id=0, accumulate=False):
        _check(self.lib.bde_gauss_draw_bwd(_ptr(g), _ptr(rho), _ptr(eps), seed, stream_id, _ptr(gmean), _ptr(grho),
                                           int(accumulate), n, _stream()), "bde_gauss_draw_bwd")

    @_on_device_of
    def gauss_kl(self, mean, rho, prior_mu, prior_sigma, n, ws, kl_out=None, gmean=None, grho=None, grad_scale=1.0,
                 grad_scale_dev=None, accumulate=False):
        _check(self.lib.bde_gauss_kl(_ptr(mean), _ptr(rho), prior_mu, prior_sigma, grad_scale, _ptr(grad_scale_dev),
                                     _ptr(gmean), _ptr(grho), int(accumulate), _ptr(kl_out), _ptr(ws), n, _stream()),
               "bde_gauss_kl")

    @_on_device_of
    def l2(self, p, l2_scale, n, ws, val_out=None, g=None, grad_scale=1.0, grad_scale_dev=None, accumulate=False):
        _check(self.lib.bde_l2(_ptr(p), l2_scale, grad_scale, _ptr(grad_scale_dev), _ptr(g), int(accumulate),
                               _ptr(val_out), _ptr(ws), n, _stream()), "bde_l2")

    @_on_device_of
    def mixture_nll(self, mean, pi, sigma1, sigma2, n, ws, val_out=None, gmean=None, grad_scale=1.0, grad_scale_dev=None,
                    accumulate=False):
        """MixturePrior "KL" (bbb.py:31-37): -sum log p(mean) and its gradient wrt the means, one pass."""
        _check(self.lib.bde_mixture_nll(_ptr(mean), pi, sigma1, sigma2, grad_scale, _ptr(grad_scale_dev), _ptr(gmean),
                                        int(accumulate), _ptr(val_out), _ptr(ws), n, _stream()), "bde_mixture_nll")

    @_on_device_of
    def local_reparam_fwd(self, mean, var, out, n, eps=None, seed=0, stream_id=0):
        _check(self.lib.bde_local_reparam_fwd(_ptr(mean), _ptr(var), _ptr(eps), seed, stream_id, _ptr(out), n, _stream()),
               "bde_local_reparam_fwd")

    @_on_device_of
    def local_reparam_bwd(self, g, var, gvar, n, eps=None, seed=0, stream_id=0):
        _check(self.lib.bde_local_reparam_bwd(_ptr(g), _ptr(var), _ptr(eps), seed, stream_id, _ptr(gvar), n, _stream()),
               "bde_local_reparam_bwd")

    @_on_device_of
    def var_operand_fwd(self, v, mode: int, out):
        """mode 0: clamp(v^2, 1e-4); 1: clamp(softplus(v)^2, 1e-4); 2: softplus(v)^2 (bbb_layers.py:66-67,71,150-153)."""
        if not (v.is_contiguous() and out.is_contiguous()):
            raise BdeKernelError("var_operand_fwd: contiguous tensors expected")
        _check(self.lib.bde_var_operand_fwd(_ptr(v, "v"), mode, _ptr(out), v.numel(), _stream()), "bde_var_operand_fwd")

    @_on_device_of
    def var_operand_bwd(self, g, v, mode: int, gv):
        if not (g.is_contiguous() and v.is_contiguous() and gv.is_contiguous()):
            raise BdeKernelError("var_operand_bwd: contiguous tensors expected")
        _check(self.lib.bde_var_operand_bwd(_ptr(g, "g"), _ptr(v), mode, _ptr(gv), v.numel(), _stream()),
               "bde_var_operand_bwd")

    @staticmethod
    def _conv_geo(x, w_shape, stride, padding):
        """(N, C, H, W, O, KH, KW, sh, sw, ph, pw), output shape; x must be a dense NCHW tensor of the layer's channels."""
        if x.dim() != 4 or len(w_shape) != 4 or not x.is_contiguous() or int(x.shape[1]) != int(w_shape[1]):
            raise BdeKernelError(f"conv_lrt: x must be a contiguous [N, C, H, W] tensor with C = {int(w_shape[1])}, got "
                                 f"{tuple(x.shape)} with strides {tuple(x.stride())}")
        n, c, h, w = (int(v) for v in x.shape)
        o, _, kh, kw = (int(v) for v in w_shape)
        sh, sw, ph, pw = int(stride[0]), int(stride[1]), int(padding[0]), int(padding[1])
        return (n, c, h, w, o, kh, kw, sh, sw, ph, pw), (n, o, (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1)

    @staticmethod
    def _dense(t, shape, name):
        if t is not None and (tuple(t.shape) != tuple(shape) or not t.is_contiguous()):
            raise BdeKernelError(f"conv_lrt: {name} must be a contiguous tensor of shape {tuple(shape)}, got {tuple(t.shape)} "
                                 f"with strides {tuple(t.stride())}")

    def conv_lrt_supported(self, x_shape, w_shape, stride, padding) -> bool:
        n, c, h, w = (int(v) for v in x_shape)
        o, c2, kh, kw = (int(v) for v in w_shape)
        return c == c2 and bool(self.lib.bde_conv_lrt_supported(n, c, h, w, o, kh, kw, stride[0], stride[1], padding[0],
                                                                 padding[1]))

    def conv_lrt_wbuf(self, w_shape, device) -> torch.Tensor:
        """The zero-initialised buffer bde_conv_lrt_prep fills (sigma^2 and the weight matrices in staging order)."""
        o, c, kh, kw = (int(v) for v in w_shape)
        return torch.zeros(int(self.lib.bde_conv_lrt_prep_floats(o, c, kh, kw)), dtype=torch.float32, device=device)

    @_on_device_of
    def conv_lrt_prep(self, w_mu, w_rho, wbuf, b_rho=None, stride=None, padding=None):
        """Once per weight version: sigma^2, its rho-derivative, the bias variance softplus(b_rho)^2 and the re-arranged
        weight matrices into ``wbuf``.  With the layer's ``stride`` / ``padding`` (bde_conv_lrt_prep_strided) also the
        per-phase input-gradient matrices of a strided layer, which ``conv_lrt_bwd_data(..., phases=True)`` needs."""
        o, c, kh, kw = (int(v) for v in w_mu.shape)
        self._dense(w_mu, (o, c, kh, kw), "w_mu")
        self._dense(w_rho, (o, c, kh, kw), "w_rho")
        self._dense(b_rho, (o,), "b_rho")
        if wbuf.numel() < int(self.lib.bde_conv_lrt_prep_floats(o, c, kh, kw)) or not wbuf.is_contiguous():
            raise BdeKernelError("conv_lrt_prep: wbuf is smaller than bde_conv_lrt_prep_floats() (use conv_lrt_wbuf)")
        if stride is not None:
            padding = (0, 0) if padding is None else padding
            _check(self.lib.bde_conv_lrt_prep_strided(_ptr(w_mu, "w_mu"), _ptr(w_rho), _ptr(b_rho), o, c, kh, kw, int(stride[0]),
                                                      int(stride[1]), int(padding[0]), int(padding[1]), _ptr(wbuf), _stream()),
                   "bde_conv_lrt_prep_strided")
            return
        _check(self.lib.bde_conv_lrt_prep(_ptr(w_mu, "w_mu"), _ptr(w_rho), _ptr(b_rho), o, c, kh, kw, _ptr(wbuf), _stream()),
               "bde_conv_lrt_prep")

    def _conv_wbuf_ok(self, wbuf, geo):
        if wbuf.numel() < int(self.lib.bde_conv_lrt_prep_floats(geo[4], geo[1], geo[5], geo[6])) or not wbuf.is_contiguous():
            raise BdeKernelError("conv_lrt: wbuf does not belong to a layer of this shape")

    @_on_device_of
    def conv_lrt_fwd(self, x, wbuf, w_shape, b_mu, bias_var, stride, padding, out, var_out, eps=None, seed=0, stream_id=0):
        """BBBConv2d forward (bbb_layers.py:146-154) in one launch; all tensors contiguous fp32 NCHW.  ``bias_var``: add
        the bias variance conv_lrt_prep evaluated from its ``b_rho``."""
        geo, oshape = self._conv_geo(x, w_shape, stride, padding)
        self._conv_wbuf_ok(wbuf, geo)
        self._dense(out, oshape, "out")
        self._dense(var_out, oshape, "var_out")            # None: a forward nobody differentiates (no variance written)
        self._dense(eps, oshape, "eps")
        self._dense(b_mu, (geo[4],), "b_mu")
        _check(self.lib.bde_conv_lrt_fwd(_ptr(x, "x"), _ptr(wbuf), _ptr(b_mu), int(bool(bias_var)), _ptr(eps), seed, stream_id,
                                         _ptr(out), _ptr(var_out), *geo, _stream()), "bde_conv_lrt_fwd")

    @_on_device_of
    def conv_lrt_bwd_data(self, g_out, g_var, wbuf, w_shape, x, g_x, stride, padding, phases=False):
        """g_x of BBBConv2d: both transposed convolutions + the clamp's derivative in one launch -- or, for a strided layer
        whose ``wbuf`` was prepared with this stride / padding (``phases=True``), one launch per phase of the output pixel
        grid: sh * sw times less matrix work than convolving the zero-dilated gradient."""
        geo, oshape = self._conv_geo(x, w_shape, stride, padding)
        self._conv_wbuf_ok(wbuf, geo)
        self._dense(g_out, oshape, "g_out")
        self._dense(g_var, oshape, "g_var")
        self._dense(g_x, tuple(x.shape), "g_x")
        fn = self.lib.bde_conv_lrt_bwd_data_phases if phases else self.lib.bde_conv_lrt_bwd_data
        _check(fn(_ptr(g_out, "g_out"), _ptr(g_var), _ptr(wbuf), _ptr(x), _ptr(g_x), *geo, _stream()),
               "bde_conv_lrt_bwd_data_phases" if phases else "bde_conv_lrt_bwd_data")

    @_on_device_of
    def conv_lrt_gvar_bias(self, g_out, var, g_var, eps=None, seed=0, stream_id=0, b_rho=None, g_bmu=None, g_brho=None):
        """First pass of BBBConv2d's backward: g_var = g eps / (2 sqrt(var)) over the layer output [N, O, Ho, Wo] (supplied
        noise, or the forward's Philox stream) and -- with ``b_rho`` -- the bias gradients g_bmu / g_brho in the same pass."""
        if g_out.dim() != 4:
            raise BdeKernelError("conv_lrt_gvar_bias: g_out must be [N, O, Ho, Wo]")
        shape = tuple(g_out.shape)
        n, o = int(shape[0]), int(shape[1])
        for t, name in ((g_out, "g_out"), (var, "var"), (g_var, "g_var"), (eps, "eps")):
            self._dense(t, shape, name)
        ws = None
        if b_rho is not None:
            for t, name in ((b_rho, "b_rho"), (g_bmu, "g_bmu"), (g_brho, "g_brho")):
                if t is None:
                    raise BdeKernelError("conv_lrt_gvar_bias: b_rho, g_bmu and g_brho come together")
                self._dense(t, (o,), name)
            ws = torch.empty(int(self.lib.bde_conv_lrt_gvar_ws_bytes(n, o)) // 8, dtype=torch.float64, device=g_out.device)
        _check(self.lib.bde_conv_lrt_gvar_bias(_ptr(g_out, "g_out"), _ptr(var), _ptr(eps), seed, stream_id, _ptr(g_var),
                                               _ptr(b_rho), _ptr(g_bmu), _ptr(g_brho), None if ws is None else _ptr64(ws, "ws"),
                                               n, o, int(shape[2]) * int(shape[3]), _stream()), "bde_conv_lrt_gvar_bias")

    @_on_device_of
    def conv_lrt_bwd_weight(self, x, g_out, g_var, w_rho, g_wmu, g_wrho, stride, padding, ws=None):
        """g_wmu / g_wrho of BBBConv2d: two weight-gradient convolutions + the rho chain rule in two launches."""
        geo, oshape = self._conv_geo(x, tuple(w_rho.shape), stride, padding)
        self._dense(g_out, oshape, "g_out")
        self._dense(g_var, oshape, "g_var")
        for t, name in ((w_rho, "w_rho"), (g_wmu, "g_wmu"), (g_wrho, "g_wrho")):
            self._dense(t, tuple(w_rho.shape), name)
        need = int(self.lib.bde_conv_lrt_bwd_weight_ws_bytes(*geo))
        if need == 0:
            raise BdeKernelError("bde_conv_lrt_bwd_weight: unsupported geometry")
        if ws is None or ws.numel() * ws.element_size() < need:
            ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
        _check(self.lib.bde_conv_lrt_bwd_weight(_ptr(x, "x"), _ptr(g_out), _ptr(g_var), _ptr(w_rho), _ptr(ws),
                                                ws.numel() * ws.element_size(), _ptr(g_wmu), _ptr(g_wrho), *geo, _stream()),
               "bde_conv_lrt_bwd_weight")
        return ws

    # ---- tuning hooks of the fused convolution (tools/conv_autotune.py, tests): candidate tilings, pinning
    @staticmethod
    def _layer_geo(x_shape, w_shape, stride, padding):
        import ctypes
        n, c, h, w = (int(v) for v in x_shape)
        o, _, kh, kw = (int(v) for v in w_shape)
        return (ctypes.c_int * 11)(n, c, h, w, o, kh, kw, int(stride[0]), int(stride[1]), int(padding[0]), int(padding[1]))

    def conv_lrt_pass_geos(self, which, x_shape, w_shape, stride, padding):
        """Launch geometries (15-tuples) of a pass of a layer: which = 0 forward, 1 dilated input gradient, 2 per-phase input
        gradient."""
        import ctypes
        out = (ctypes.c_int * (15 * 64))()
        n = int(self.lib.bde_conv_lrt_pass_geos(int(which), self._layer_geo(x_shape, w_shape, stride, padding), out, 64))
        if n < 0:
            raise BdeKernelError("bde_conv_lrt_pass_geos: unsupported layer geometry")
        return [tuple(out[15 * i:15 * i + 15]) for i in range(min(n, 64))]

    def conv_lrt_candidates(self, geo):
        """([(WK, TH, NI, CC, PT, LDS bytes), ...], index the planner runs) for a launch geometry."""
        import ctypes
        g = (ctypes.c_int * 15)(*[int(v) for v in geo])
        out, chosen = (ctypes.c_int * (6 * 512))(), ctypes.c_int(-1)
        n = int(self.lib.bde_conv_lrt_candidates(g, out, 512, ctypes.byref(chosen)))
        if n < 0:
            raise BdeKernelError("bde_conv_lrt_candidates: invalid launch geometry")
        return [tuple(out[6 * i:6 * i + 6]) for i in range(min(n, 512))], int(chosen.value)

    def conv_lrt_set_tiling(self, geo, tiling=None):
        """Pin (WK, TH, NI, CC) for a launch geometry (None: remove the pin)."""
        import ctypes
        g = (ctypes.c_int * 15)(*[int(v) for v in geo])
        wk, th, ni, cc = (0, 0, 0, 0) if tiling is None else (int(v) for v in tiling[:4])
        _check(self.lib.bde_conv_lrt_set_tiling(g, wk, th, ni, cc), "bde_conv_lrt_set_tiling")

    def conv_lrt_wgrad_candidates(self, x_shape, w_shape, stride, padding):
        """([(CT, TH, NI, PS, LDS bytes), ...], index the planner runs) for the weight-gradient pass of a layer."""
        import ctypes
        out, chosen = (ctypes.c_int * (5 * 512))(), ctypes.c_int(-1)
        n = int(self.lib.bde_conv_lrt_wgrad_candidates(self._layer_geo(x_shape, w_shape, stride, padding), out, 512,
                                                       ctypes.byref(chosen)))
        if n < 0:
            raise BdeKernelError("bde_conv_lrt_wgrad_candidates: invalid layer geometry")
        return [tuple(out[5 * i:5 * i + 5]) for i in range(min(n, 512))], int(chosen.value)

    def conv_lrt_wgrad_set_tiling(self, x_shape, w_shape, stride, padding, tiling=None):
        """Pin (CT, TH, NI, PS) for a layer's weight-gradient pass (None: remove the pin).  Partials buffers sized before the
        pin must not be reused (conv_lrt_bwd_weight re-checks the size)."""
        ct, th, ni, ps = (0, 0, 0, 0) if tiling is None else (int(v) for v in tiling[:4])
        _check(self.lib.bde_conv_lrt_wgrad_set_tiling(self._layer_geo(x_shape, w_shape, stride, padding), ct, th, ni, ps),
               "bde_conv_lrt_wgrad_set_tiling")

    def lrt_linear_supported(self, b: int, i: int, o: int) -> bool:
        return bool(self.lib.bde_lrt_linear_supported(b, i, o))

    def lrt_sigma_cache_wanted(self, i: int, o: int) -> bool:
        """True for layers wide enough that the kernels read a per-weight-version sigma^2 cache instead of evaluating
        softplus / sigmoid per weight per pass."""
        return bool(self.lib.bde_lrt_sigma_cache_wanted(i, o))

    @_on_device_of
    def lrt_sigma_cache(self, w_rho, s2, ds2=None):
        """s2 = clamp(softplus(rho)^2, 1e-4), ds2 = [sigma^2 >= 1e-4] * 2 sigma sigmoid(rho): one pass per weight version."""
        if not (w_rho.is_contiguous() and s2.is_contiguous() and (ds2 is None or ds2.is_contiguous())):
            raise BdeKernelError("lrt_sigma_cache: contiguous tensors expected")
        _check(self.lib.bde_lrt_sigma_cache(_ptr(w_rho, "w_rho"), _ptr(s2), _ptr(ds2), w_rho.numel(), _stream()),
               "bde_lrt_sigma_cache")

    @_on_device_of
    def lrt_linear_fwd(self, x, w_mu, w_rho, b_mu, b_rho, clamp_bias_var, out, var_out, eps=None, seed=0, stream_id=0,
                       w_s2=None):
        """Fused local-reparameterisation forward of a mean-field linear layer (bbb_layers.py:61-80): x [B, I] (row
        stride free), w_mu / w_rho [O, I] contiguous, out / var_out / eps [B, O] contiguous."""
        b, i = x.shape
        o = w_mu.shape[0]
        if not (w_mu.is_contiguous() and w_rho.is_contiguous() and out.is_contiguous()):
            raise BdeKernelError("w_mu, w_rho and out must be contiguous")
        n = self.lib.bde_lrt_linear_ws_bytes(b, i, o)
        if n == 0:
            raise BdeKernelError(f"lrt_linear_fwd: unsupported shape B={b}, I={i}, O={o}")
        ws = torch.empty(n // 4, dtype=torch.float32, device=x.device)
        _check(self.lib.bde_lrt_linear_fwd(_ptr(x, "x"), x.stride(0), _ptr(w_mu), _ptr(w_rho), _ptr(w_s2), _ptr(b_mu),
                                           _ptr(b_rho), int(clamp_bias_var), _ptr(eps), seed, stream_id, _ptr(out),
                                           _ptr(var_out), b, i, o, _ptr(ws), _stream()), "bde_lrt_linear_fwd")

    @_on_device_of
    def lrt_linear_bwd(self, x, w_mu, w_rho, b_rho, clamp_bias_var, g, var, g_x, g_wmu, g_wrho, g_bmu, g_brho, eps=None,
                       seed=0, stream_id=0, w_s2=None, w_ds2=None):
        """Backward of lrt_linear_fwd (the autograd graph of bbb_layers.py:61-80): g / var / eps [B, O] contiguous,
        g_x [B, I] (None: not wanted), g_wmu / g_wrho [O, I], g_bmu / g_brho [O] (None with b_rho None); all
        outputs are overwritten."""
        b, i = x.shape
        o = w_mu.shape[0]
        for t in (w_mu, w_rho, g, var, g_x, g_wmu, g_wrho, eps):
            if t is not None and not t.is_contiguous():
                raise BdeKernelError("lrt_linear_bwd: weights, g, var, eps and the gradient outputs must be contiguous")
        n = self.lib.bde_lrt_linear_bwd_ws_bytes(b, i, o)
        if n == 0:
            raise BdeKernelError(f"lrt_linear_bwd: unsupported shape B={b}, I={i}, O={o}")
        ws = torch.empty(n // 4, dtype=torch.float32, device=x.device)
        _check(self.lib.bde_lrt_linear_bwd(_ptr(x, "x"), x.stride(0), _ptr(w_mu), _ptr(w_rho), _ptr(w_s2), _ptr(w_ds2),
                                           _ptr(b_rho), int(clamp_bias_var), _ptr(g), _ptr(var), _ptr(eps), seed, stream_id, _ptr(g_x),
                                           _ptr(g_wmu), _ptr(g_wrho), _ptr(g_bmu), _ptr(g_brho), b, i, o, _ptr(ws),
                                           _stream()), "bde_lrt_linear_bwd")

    # ------------------------------------------------------------ iVON --
    @_on_device_of
    def ivon_sample(self, mean, prec, param, delta_sum, n, n_eff, first, eps=None, seed=0, stream_id=0,
                    deterministic=False):
        _check(self.lib.bde_ivon_sample(_ptr(mean), _ptr(prec), _ptr(eps), seed, stream_id, n_eff, int(deterministic),
                                        int(first), _ptr(param), _ptr(delta_sum), n, _stream()), "bde_ivon_sample")

    @_on_device_of
    def ivon_update(self, mean, momentum, prec, delta_sum, acc_grad, n, *, lam, n_eff, mc, beta1, beta2, t, lr,
                    damping):
        # Python-double scalar expressions exactly as ivorn.py:72-89 forms them
        _check(self.lib.bde_ivon_update(_ptr(mean), _ptr(momentum), _ptr(prec), _ptr(delta_sum), _ptr(acc_grad),
                                        lam, n_eff, mc, beta1, 1 - beta1, 1 - beta2, 0.5 * (1 - beta2) ** 2,
                                        1 - beta1 ** t, 1 - beta2 ** t, lr, damping, n, _stream()), "bde_ivon_update")
