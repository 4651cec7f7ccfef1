#!/usr/bin/env python3
"""Headline benchmark: SVGD posterior-update steps/s (+ SWAG samples/s and the
other hot-path kernels as extras) on ResNet-50-sized flat weight buffers.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one SVGD posterior update (src/algos/svgd.py:83-89 of the
reference) of M = 8 particles with D = 23,880,950 parameters each (the
reference's iWildCam ResNet-50), from P and G resident in
HBM to -phi: three launches (MFMA Gram, kernel statistics, streaming combine),
16*M*D algorithmic bytes.  Model forward/backward is not part of the path.

N > 1: the 8 particles are sharded M/N per rank and a step is the product's own
multi-GPU posterior update (SVGDOptimizer._posterior_update, fused SGD base
optimizer): gradient exchange over RCCL/xGMI + update, total work fixed
("strong").  ONE invocation times every exchange mode, each with a freshly
built optimizer: "allgather" (one all-gather of the gradient rows, then the
update -- the exchange north_star names and the headline value), "pipelined"
(chunked all-gather overlapped with the update) and "alltoall"
(dimension-sharded); `exchange` in the JSON carries all three and names the
best.  A failing collective ends the run non-zero (no in-process fallback).

Timing: after W warm-up steps, --blocks (default 5) blocks of EXACTLY K steps,
each bracketed by barrier + synchronize and reduced with MAX over ranks;
ms_per_step is the MEDIAN block (all blocks are reported).

One JSON line on stdout (rank 0); progress goes to stderr.

Processes.  N = 1 (the default): the process started here NEVER touches the GPU.  It runs the parts as children, one
after the other -- `--headline-child` (timed region, roofline with HIP-event launch times and the in-run probe, SWAG
rates, the reference's op sequence on the same GPU; under torchrun with one rank also the forced RCCL exchange),
`--extras-child` (every other kernel, the shells, the other BASELINE configs) and two `rocprofv3 --pmc ... -- python3
bench.py --traffic-child` passes for roofline.traffic -- times the CPU baseline itself on the host cores, merges what
the children wrote and prints the line.  Every part has a time limit and a failure record of its own (`error`
beside what was measured); only a missing headline makes the exit code non-zero.  N > 1, `--extras-in-process` and
runs under rocprofv3 (whose preloaded library has initialised the GPU already) measure everything in the one
process, with every optional section guarded the same way.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

M = 8
D_RESNET50 = 23_880_950     # torchvision ResNet-50 backbone + 182-class head (iWildCam), SURVEY.md section 8
D_RESNET20 = 273_610        # CIFAR ResNet-20 (swish/FRN)
D_DENSENET = 6_955_906      # Camelyon DenseNet-121
K_SWAG, S_SWAG = 20, 30
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
DATASET_SIZE = 129_809.0    # iwildcam.yaml:217-221


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def pad_ld(d):
    return (d + 16 + 63) // 64 * 64


def make_svgd_inputs(d, dev, seed, shared_backbone=True):
    """P = theta0 + head-only perturbation (mimics iwildcam/models.py:118-119: particles
    share the pretrained backbone, only the head is re-initialised); G ~ N(0, 0.01^2)."""
    ld = pad_ld(d)
    g = torch.Generator(device=dev).manual_seed(seed)
    P = torch.zeros(M, ld, device=dev)
    if shared_backbone:
        theta0 = torch.randn(d, device=dev, generator=g) * 0.05
        P[:, :d] = theta0
        head = min(d, 372_918)
        P[:, d - head:d] += (torch.rand(M, head, device=dev, generator=g) * 2 - 1) / (2048 ** 0.5)
    else:
        P[:, :d] = torch.randn(M, d, device=dev, generator=g) * 0.05
    G = torch.zeros(M, ld, device=dev)
    G[:, :d] = torch.randn(M, d, device=dev, generator=g) * 0.01
    return P, G


def time_loop(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


class StreamProbes:
    """bench_probe/libbde_bench_probe.so (bench-only): a no-arithmetic kernel with the read / write / read-modify-write
    SHAPE of a product kernel, run on the SAME tensors right beside it.  frac_of_probe = kernel rate / probe rate."""

    def __init__(self):
        import ctypes
        self.lib = None
        path = os.path.join(ROOT, "bench_probe", "libbde_bench_probe.so")
        if os.path.exists(path):
            self.lib = ctypes.CDLL(path)
            c = ctypes
            self.lib.bde_bench_probe.restype = c.c_int
            self.lib.bde_bench_probe.argtypes = [c.c_int, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_int64, c.c_int, c.c_int64,
                                                 c.c_void_p, c.c_int64, c.c_int, c.c_int64, c.c_void_p, c.c_int64, c.c_int64,
                                                 c.c_void_p]

    @staticmethod
    def _rows(tensors):
        """(base pointer, floats between consecutive rows) of 1..n row tensors: one 2-D tensor, or separately allocated
        vectors (then the distance between the first two; any further ones are assumed equally spaced)."""
        if tensors is None:
            return 0, 0
        if torch.is_tensor(tensors):
            return tensors.data_ptr(), (tensors.stride(0) if tensors.dim() == 2 else 0)
        ptrs = [t.data_ptr() for t in tensors]
        step = (ptrs[1] - ptrs[0]) // 4 if len(ptrs) > 1 else 0
        assert all(b - a == 4 * step for a, b in zip(ptrs, ptrs[1:])), "probe rows must be equally spaced"
        return ptrs[0], step

    def run(self, shape, n, rd=None, wr=None, rw=None, nt_store=False, rd_pieces=None, wr_pieces=None):
        """One launch of the (n_read, n_write, n_rmw) probe over n floats per row."""
        rp, rl = self._rows(rd)
        wp, wl = self._rows(wr)
        mp, ml = self._rows(rw)
        rlp, rps = rd_pieces if rd_pieces else (0, 0)          # (log2_piece, piece_stride): rows interleaved in pieces
        wlp, wps = wr_pieces if wr_pieces else (0, 0)
        rc = self.lib.bde_bench_probe(shape[0], shape[1], shape[2], int(nt_store), rp, rl, rlp, rps, wp, wl, wlp, wps, mp, ml, n,
                                      torch.cuda.current_stream().cuda_stream)
        assert rc == 0, (shape, rc)


def extras(ops, dev, quick, on_section=None):
    """Secondary hot-path kernels: seconds per launch, algorithmic GB/s, fraction of the 8 TB/s HBM peak, and -- round 4 --
    beside every streaming kernel a no-arithmetic probe of the same read / write shape on the same tensors
    (probe_GBps, frac_of_probe = kernel rate / probe rate; StreamProbes)."""
    out = {}
    it = 10 if quick else 20
    probes = StreamProbes()

    def rec(name, t, nbytes, unit_count=None, unit=None, probe=None, probe_iters=None):
        e = {"ms": round(t * 1e3, 4), "GBps": round(nbytes / t / 1e9, 1), "hbm_frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4)}
        if unit_count is not None:
            e[unit] = round(unit_count / t, 1)
        if probe is not None and probes.lib is not None:
            try:                                     # a probe is bench-only garnish: it must never cost the bench line
                shape, n = probe["shape"], probe["n"]
                assert 4 * n * (shape[0] + shape[1] + 2 * shape[2]) == nbytes, (name, shape, nbytes)
                kw = {k: v for k, v in probe.items() if k not in ("shape", "n")}
                tp = time_loop(lambda: probes.run(shape, n, **kw), probe_iters or it)
                e["probe_shape"] = "R%d W%d RMW%d" % shape
                e["probe_GBps"] = round(nbytes / tp / 1e9, 1)
                e["frac_of_probe"] = round(tp / t, 4)
            except Exception as err:
                e["probe_error"] = f"{type(err).__name__}: {err}"
        out[name] = e
        log(f"  {name:34s} {t*1e3:9.3f} ms {nbytes/t/1e9:8.1f} GB/s {e['hbm_frac']*100:5.1f}%"
            + (f"   probe {e['probe_GBps']:8.1f} GB/s  of probe {e['frac_of_probe']:.3f}" if "frac_of_probe" in e else ""))

    d = D_RESNET50
    ld = pad_ld(d)
    g = torch.Generator(device=dev).manual_seed(99)
    # --- SVGD variants
    P, G = make_svgd_inputs(d, dev, 1234)
    outb = torch.empty_like(G)
    ws, ks = ops.svgd_ws(M, dev), ops.svgd_kstat(M, dev)
    rec("svgd_gram_M8_resnet50", time_loop(lambda: ops.svgd_gram(P, d, ws), it), 4 * M * d, probe=dict(shape=(8, 0, 0), n=d, rd=P))
    rec("svgd_combine_M8_resnet50", time_loop(lambda: ops.svgd_combine(P, G, outb, d, ks), it), 12 * M * d)
    rec("svgd_combine_inplace_M8_resnet50", time_loop(lambda: ops.svgd_combine(P, outb, outb, d, ks), it), 12 * M * d)
    buf = torch.zeros(ld, device=dev)
    t = time_loop(lambda: (ops.svgd_step(P, G, outb, d, 0.0, 1.0, DATASET_SIZE, -1.0, ws, ks),
                           ops.svgd_apply_sgd(P, outb, buf, d, 1e-12, 0.9, 0.0, 3e-4, True, False)), it)
    rec("svgd_step_then_apply_sgd_M8_resnet50", t, (16 * M + 12 * M + 8) * d, 1, "steps_per_s")
    # whole SVGDOptimizer.step minus forward/backward, fused: kernel stats + ONE pass (-phi in registers, M
    # shared-state SGD applications, updated particles out, Gram partials of the updated particles for the next step)
    wsn = ops.svgd_ws(M, dev)
    ops.svgd_gram(P, d, wsn)
    t = time_loop(lambda: (ops.svgd_kstats(wsn, M, 0.0, 1.0, DATASET_SIZE, -1.0, ks),
                           ops.svgd_fused_sgd(P, G, buf, d, ks, 1e-12, 0.9, 0.0, 3e-4, True, False, ws_next=wsn)), it)
    rec("svgd_full_step_fused_sgd_reuse_gram_M8_resnet50", t, (12 * M + 8) * d, 1, "steps_per_s")
    Pi, Gi = make_svgd_inputs(d, dev, 1234, shared_backbone=False)
    rec("svgd_step_M8_resnet50_independent_particles",
        time_loop(lambda: ops.svgd_step(Pi, Gi, outb, d, 0.0, 1.0, DATASET_SIZE, -1.0, ws, ks), it), 16 * M * d, 1, "steps_per_s")
    del Pi, Gi, P, G, outb
    # --- SWAG: the K + 2 statistics rows are the rows of one [K + 2, ld] buffer (ring rows, then mean, then second moment)
    stat = torch.randn(K_SWAG + 2, ld, device=dev, generator=g) * 1e-3
    stat[K_SWAG] = torch.randn(ld, device=dev, generator=g) * 0.05
    stat[K_SWAG + 1] = stat[K_SWAG] * stat[K_SWAG] + 1e-4
    ring, mean, sq = stat[:K_SWAG], stat[K_SWAG], stat[K_SWAG + 1]
    theta = torch.randn(ld, device=dev, generator=g) * 0.05
    o = torch.empty(ld, device=dev)
    rec("swag_update_resnet50", time_loop(lambda: ops.swag_update(theta, mean, sq, ring[3], 5, d), it), 24 * d,
        probe=dict(shape=(1, 1, 2), n=d, rd=theta, wr=ring[3], rw=[mean, sq]))
    t = time_loop(lambda: ops.swag_sample(mean, sq, ring, 3, o, d, seed=1, stream_id=2), it)
    rec("swag_sample_K20_resnet50", t, 4 * d * (K_SWAG + 3), 1, "samples_per_s", probe=dict(shape=(22, 1, 0), n=d, rd=stat, wr=o))
    # batched sampler and its same-shape probe INTERLEAVED (5 rounds of 6 launches each, same allocation state); beside
    # them the probe on round 3's layout (rows interleaved in 16 KB pieces, [piece][row][4096 floats]): what the
    # silicon gives each layout without any arithmetic -- the layout question of VERDICT r3 #3, in driver-visible data
    ob = torch.empty(S_SWAG, ld, device=dev)
    nb = 4 * d * (K_SWAG + 2 + S_SWAG)
    arms = {"kernel": lambda: ops.swag_sample_batched(mean, sq, ring, 3, ob, d, seed=1, stream_id0=0)}
    if probes.lib is not None and hasattr(probes.lib, "bde_bench_probe"):
        npc = (d + 4095) // 4096
        pin = torch.zeros(npc, K_SWAG + 2, 4096, device=dev)
        pout = torch.empty(npc, S_SWAG, 4096, device=dev)
        arms["probe"] = lambda: probes.run((22, 30, 0), d, rd=stat, wr=ob, nt_store=True)
        arms["probe_rows_in_pieces"] = lambda: probes.run((22, 30, 0), d, rd=pin[0], wr=pout[0], nt_store=True,
                                                          rd_pieces=(12, (K_SWAG + 2) * 4096), wr_pieces=(12, S_SWAG * 4096))
    for fn in arms.values():
        time_loop(fn, 3)
    times = {k: [] for k in arms}
    for _ in range(3 if quick else 5):
        for k, fn in arms.items():
            times[k].append(time_loop(fn, 6, warm=1))
    ts = sorted(times["kernel"])
    rec("swag_sample_batched_K20_S30_resnet50", ts[len(ts) // 2], nb, S_SWAG, "samples_per_s")
    e = out["swag_sample_batched_K20_S30_resnet50"]
    e["ms_min"], e["ms_all_rounds"] = round(ts[0] * 1e3, 4), [round(x * 1e3, 4) for x in times["kernel"]]
    e["hbm_frac_best_round"] = round(nb / ts[0] / 1e9 / HBM_PEAK_GBS, 4)
    e["timing"] = "median of 5 interleaved rounds (kernel, probe, probe on rows in pieces; 6 launches each)"
    if "probe" in times:
        tp, tq = sorted(times["probe"]), sorted(times["probe_rows_in_pieces"])
        e["probe_shape"], e["probe_GBps"] = "R22 W30 RMW0", round(nb / tp[len(tp) // 2] / 1e9, 1)
        e["frac_of_probe"] = round(tp[len(tp) // 2] / ts[len(ts) // 2], 4)
        e["probe_rows_in_pieces_GBps"] = round(nb / tq[len(tq) // 2] / 1e9, 1)
        del pin, pout
    rec("swag_serve_prefetched_sample_resnet50", time_loop(lambda: o.copy_(ob[7]), it), 8 * d, 1, "samples_per_s",
        probe=dict(shape=(1, 1, 0), n=d, rd=ob[7], wr=o))
    del ob, ring, stat
    # --- BBB
    mean = torch.randn(ld, device=dev, generator=g) * 0.05
    rho = torch.full((ld,), -3.0, device=dev)
    w = torch.empty(ld, device=dev)
    gm, gr = torch.zeros(ld, device=dev), torch.zeros(ld, device=dev)
    rws, kl = ops.reduce_ws(dev), torch.zeros(1, device=dev)
    rec("bbb_draw_fwd_resnet50", time_loop(lambda: ops.gauss_draw_fwd(mean, rho, w, d, seed=1, stream_id=0), it), 12 * d,
        probe=dict(shape=(2, 1, 0), n=d, rd=[mean, rho], wr=w))
    rec("bbb_draw_bwd_resnet50", time_loop(lambda: ops.gauss_draw_bwd(w, rho, gm, gr, d, seed=1, stream_id=0, accumulate=True), it), 24 * d,
        probe=dict(shape=(2, 0, 2), n=d, rd=[w, rho], rw=[gm, gr]))
    # the KL row is the ACCUMULATE form (mu, rho read; gmu, grho read-modify-write) = 24 B/param
    rec("bbb_kl_fwd_bwd_resnet50", time_loop(lambda: ops.gauss_kl(mean, rho, 0.0, 1.0, d, rws, kl_out=kl, gmean=gm, grho=gr, accumulate=True), it), 24 * d,
        probe=dict(shape=(2, 0, 2), n=d, rd=[mean, rho], rw=[gm, gr]))
    rec("bbb_kl_fwd_bwd_overwrite_resnet50", time_loop(lambda: ops.gauss_kl(mean, rho, 0.0, 1.0, d, rws, kl_out=kl, gmean=gm, grho=gr), it), 16 * d,
        probe=dict(shape=(2, 2, 0), n=d, rd=[mean, rho], wr=[gm, gr]))
    var = torch.rand(ld, device=dev) + 1e-4
    rec("bbb_local_reparam_epilogue_fwd_resnet50", time_loop(lambda: ops.local_reparam_fwd(mean, var, w, d, seed=1, stream_id=0), it), 12 * d,
        probe=dict(shape=(2, 1, 0), n=d, rd=[mean, var], wr=w))
    # --- iVON
    prec = torch.full((ld,), 100.0 / DATASET_SIZE, device=dev)
    ds, mom = torch.zeros(ld, device=dev), torch.zeros(ld, device=dev)
    rec("ivon_sample_resnet50", time_loop(lambda: ops.ivon_sample(mean, prec, w, ds, d, DATASET_SIZE, first=False, seed=1, stream_id=0), it), 20 * d,
        probe=dict(shape=(2, 1, 1), n=d, rd=[mean, prec], wr=w, rw=ds))
    st3 = torch.zeros(3, ld, device=dev)                     # mean / momentum / precision as rows of one tensor (probe: equal spacing)
    st3[0], st3[2] = mean, prec
    mean2, mom, prec = st3[0], st3[1], st3[2]
    rec("ivon_update_resnet50", time_loop(lambda: ops.ivon_update(mean2, mom, prec, ds, gm, d, lam=100.0 / DATASET_SIZE, n_eff=DATASET_SIZE, mc=2,
                                                                   beta1=0.9, beta2=0.999, t=1, lr=1e-12, damping=1e-3), it), 32 * d,
        probe=dict(shape=(2, 0, 3), n=d, rd=[ds, gm], rw=[mean2, mom, prec]))
    if on_section is not None:
        on_section(out)
    # BASELINE configs[1] size (CIFAR ResNet-20, D = 273,610) with the kernels a default call runs: the three streaming launches
    d20 = D_RESNET20
    P2, G2 = make_svgd_inputs(d20, dev, 1234)
    o2 = torch.empty_like(G2)
    b2 = torch.zeros(pad_ld(d20), device=dev)
    rec("svgd_step_M8_resnet20", time_loop(lambda: ops.svgd_step(P2, G2, o2, d20, 3e-4, 1.0, 50000.0, -1.0, ws, ks), 50),
        16 * M * d20, 1, "steps_per_s")

    def streaming_fused():
        ops.svgd_gram(P2, d20, ws)
        ops.svgd_kstats(ws, M, 3e-4, 1.0, 50000.0, -1.0, ks)
        ops.svgd_fused_sgd(P2, G2, b2, d20, ks, 1e-12, 0.9, 0.0, 3e-4, True, False)
    rec("svgd_full_step_fused_sgd_M8_resnet20_streaming", time_loop(streaming_fused, 50), (16 * M + 8) * d20, 1, "steps_per_s")
    if on_section is not None:
        on_section(out)
    # LAST: the small-model kernel was rewritten after its last run on an MI355X and is no default until
    # its parity tests have been green on a device (device_verified.py); what was measured above has been handed to
    # `on_section` before it is launched, and each of its entries fails on its own
    for name, fn, nbytes in (
            ("svgd_step_M8_resnet20_small_kernel",
             lambda: ops.svgd_step_small(P2, G2, o2, d20, 3e-4, 1.0, 50000.0, -1.0, ws, ks), 16 * M * d20),
            ("svgd_full_step_fused_sgd_M8_resnet20_2_launches",
             lambda: ops.svgd_step_small_sgd(P2, G2, b2, d20, 3e-4, 1.0, 50000.0, ws, ks, 1e-12, 0.9, 0.0, 3e-4, True, False),
             (12 * M + 8) * d20)):
        try:
            rec(name, time_loop(fn, 50), nbytes, 1, "steps_per_s")
        except Exception as e:                                       # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    del P2, G2, o2, b2
    out["probe_note"] = ("probe_GBps / frac_of_probe: a no-arithmetic kernel with the SAME number of read / written / "
                         "read-modify-written rows on the SAME tensors (bench_probe/probe.hip), timed right beside the kernel: "
                         "frac_of_probe = kernel rate / probe rate; the probe walks its rows at the kernel's own addresses")
    return out


def stream_probe(ops, P, G, out, d, ws, ks, iters=20):
    """The same-shape HBM stream probe (bench_probe/probe.hip, a bench-only library): 16 rows read + 8 rows written with
    the combine kernel's walk and load/store flavours but no arithmetic.  Timed with HIP events on the launch stream,
    (a) in the step's surroundings -- right behind a Gram pass over the same particle rows, as the combine kernel runs
    (so both see the same Infinity-Cache state) -- and (b) back to back; the combine kernel is re-timed back to back
    beside it.  roofline.frac_of_probe = in-step kernel rate / in-step probe rate: a device-independent figure."""
    import ctypes
    path = os.path.join(ROOT, "bench_probe", "libbde_bench_probe.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    fn = lib.bde_bench_probe_r16w8
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    ld = P.stride(0)
    stream = torch.cuda.current_stream().cuda_stream

    def probe():
        rc = fn(P.data_ptr(), G.data_ptr(), out.data_ptr(), ld, d, stream)
        assert rc == 0, rc
    nbytes = 12 * M * d

    def timed(body, pre=None):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for _ in range(3):
            if pre:
                pre()
            body()
        torch.cuda.synchronize()
        for a, b in ev:
            if pre:
                pre()
            a.record()
            body()
            b.record()
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / iters

    def gram_and_stats():
        ops.svgd_gram(P, d, ws)
        ops.svgd_kstats(ws, M, 0.0, 1.0, DATASET_SIZE, -1.0, ks)
    t_in = timed(probe, pre=gram_and_stats)
    t_b2b = timed(probe)
    t_comb_b2b = timed(lambda: ops.svgd_combine(P, G, out, d, ks))
    return {"probe_in_step_ms": t_in, "probe_back_to_back_ms": t_b2b, "combine_back_to_back_ms": t_comb_b2b,
            "probe_in_step_GBps": nbytes / (t_in * 1e-3) / 1e9, "probe_back_to_back_GBps": nbytes / (t_b2b * 1e-3) / 1e9,
            "combine_back_to_back_GBps": nbytes / (t_comb_b2b * 1e-3) / 1e9}


def cpu_baseline(P, G, d, budget_s=25.0):
    """The reference's CPU path for the same step: the oracle's torch-CPU restatement
    (same ATen op sequence as svgd.py:86-89) on the host cores, same inputs."""
    from oracle import bde_oracle as O
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    Pc, Gc = P[:, :d].cpu().contiguous(), G[:, :d].cpu().contiguous()
    t0 = time.perf_counter()
    O.cpu_svgd_step(Pc, Gc, 0.0, 1.0, DATASET_SIZE)             # warm-up
    warm = time.perf_counter() - t0
    reps = int(max(1, min(5, (budget_s - warm) // max(warm, 1e-3))))
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        O.cpu_svgd_step(Pc, Gc, 0.0, 1.0, DATASET_SIZE)
        best = min(best, time.perf_counter() - t0)
    out = {"value": round(1.0 / best, 4), "unit": "steps/s", "cores": cores, "kind": "port",
           "sample": f"the full workload (M=8, D={d}): 1 warm-up + min of {reps} timed steps of oracle.cpu_svgd_step, "
                     f"torch {torch.__version__} CPU, {cores} threads", "ms_per_step": round(best * 1e3, 2)}
    # the second half of the metric: one SWAG posterior sample the reference's way (swag.py:57,107-114: the cached
    # LowRankMultivariateNormal built once, then .sample()), K = 20, same D, on the same host cores
    try:
        del Pc, Gc
        g = torch.Generator().manual_seed(5)
        mean = torch.randn(d, generator=g) * 0.05
        sq = mean * mean + 1e-4
        devs = torch.randn(d, K_SWAG, generator=g) * 1e-3
        t0 = time.perf_counter()
        dist_obj = O.swag_build_dist(mean, sq, devs)
        build_s = time.perf_counter() - t0
        dist_obj.sample()
        best_s = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            dist_obj.sample()
            best_s = min(best_s, time.perf_counter() - t0)
        out["swag"] = {"samples_per_s": round(1.0 / best_s, 3), "ms_per_sample": round(best_s * 1e3, 2),
                       "build_distribution_ms": round(build_s * 1e3, 1), "K": K_SWAG,
                       "sample": "min of 3 LowRankMultivariateNormal.sample() calls after 1 warm-up; the distribution "
                                 "object (built once per SWAG update in the reference) is timed separately"}
    except Exception as e:                       # informational
        out["swag"] = {"skipped": f"{type(e).__name__}: {e}"}
    return out


def torch_gpu_baseline(P, G, d, dev):
    """The reference's own op sequence (oracle functions = same ATen calls as svgd.py:86-89 and
    swag.py:57,112-114) executed by PyTorch-ROCm ON THE SAME GPU: what a user of the reference gets on an
    MI355X without this library.  Reported next to the HIP numbers; not a target."""
    from oracle import bde_oracle as O
    out = {}
    Pd, Gd = P[:, :d].contiguous(), G[:, :d].contiguous()
    t = time_loop(lambda: O.cpu_svgd_step(Pd, Gd, 0.0, 1.0, DATASET_SIZE), 5, warm=2)
    out["svgd_step_ms"] = round(t * 1e3, 3)
    out["svgd_steps_per_s"] = round(1.0 / t, 2)
    del Pd, Gd
    g = torch.Generator(device=dev).manual_seed(5)
    mean = torch.randn(d, device=dev, generator=g) * 0.05
    sq = mean * mean + 1e-4
    devs = torch.randn(d, K_SWAG, device=dev, generator=g) * 1e-3
    dist_obj = O.swag_build_dist(mean, sq, devs)                   # swag.py:107-114 (cached between updates)
    t = time_loop(lambda: dist_obj.sample(), 5, warm=2)            # swag.py:57
    out["swag_sample_ms"] = round(t * 1e3, 3)
    out["swag_samples_per_s"] = round(1.0 / t, 2)
    out["what"] = "oracle functions (the reference's ATen op sequence) on CUDA tensors, torch " + torch.__version__
    return out


def shell_step_ms(dev, steps=10, n_tensors=161, d=D_RESNET50):
    """End-to-end SVGDOptimizer.step() over 161 parameter tensors totalling ResNet-50 size with NULL closures
    (forward returns a constant, backward does nothing): what is timed is the shell's host logic (re-pointing the
    views, gradient hand-over) + the kernels + the fused base optimizer -- "step minus closures" with nothing to
    subtract.  tools/shell_bench.py has the variants (unfused, Adam, closures with real gradients)."""
    import beyond_deep_ensembles_amd as bde
    sizes = [d // n_tensors] * (n_tensors - 1)
    sizes.append(d - sum(sizes))
    params = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.05) for s in sizes]
    base = torch.optim.SGD(params, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=3e-4)

    def reset():
        with torch.no_grad():
            for p in params[-2:]:
                p.normal_(0, 0.05)
    opt = bde.SVGDOptimizer(params, reset, base, particle_count=M, dataset_size=DATASET_SIZE, fuse_base_optimizer=True,
                            reuse_gram=True)
    zero = torch.zeros((), device=dev)
    fwd, bwd = (lambda: zero), (lambda loss: None)
    for _ in range(3):
        opt.step(fwd, bwd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.step(fwd, bwd)
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / steps
    from beyond_deep_ensembles_amd import _host
    return {"shell_plus_kernels_ms": round(t_step * 1e3, 3), "tensors": n_tensors, "particles": M,
            "native_host_helper": _host.load() is not None,
            "what": "SVGDOptimizer(fuse_base_optimizer=True, reuse_gram=True).step with null closures: host logic "
                    f"of the shell + kernels + fused SGD for 8 particles x {n_tensors} tensors, D = {d}"}


def resnet50_shapes(n_classes=182):
    """The 161 parameter tensors of the reference's iWildCam model (torchvision ResNet-50 + a 182-class head,
    experiments/iwildcam/models.py:21-22,170-176) in parameters() order, written out analytically (torchvision is not
    installed here): 23,880,950 elements, from 64-element batch-norm vectors to the 2.36 M-element 3x3 convolutions."""
    shapes = [(64, 3, 7, 7), (64,), (64,)]
    inpl = 64
    for planes, blocks in ((64, 3), (128, 4), (256, 6), (512, 3)):
        for b in range(blocks):
            shapes += [(planes, inpl, 1, 1), (planes,), (planes,), (planes, planes, 3, 3), (planes,), (planes,),
                       (4 * planes, planes, 1, 1), (4 * planes,), (4 * planes,)]
            if b == 0:
                shapes += [(4 * planes, inpl, 1, 1), (4 * planes,), (4 * planes,)]
            inpl = 4 * planes
    return shapes + [(n_classes, 2048), (n_classes,)]


class _ManyGrads(torch.autograd.Function):
    """loss = sum_i <p_i, c_i> as ONE autograd node with n_tensors inputs whose backward hands out n_tensors FRESH,
    separately allocated gradient tensors (c_i * grad_out through one multi-tensor launch) -- what a real model's
    backward produces, without a model's cost."""

    @staticmethod
    def forward(ctx, cs, *params):
        ctx.cs = cs
        return torch.stack([torch.dot(p.detach().view(-1)[:64], c.view(-1)[:64]) for p, c in zip(params[:2], cs[:2])]).sum()

    @staticmethod
    def backward(ctx, grad_out):
        return (None,) + tuple(torch._foreach_mul(ctx.cs, grad_out))


def shell_step_real_grads_ms(dev, n_tensors=161, d=D_RESNET50, steps=20, fuse=True, ctor="opt_in", particles=M):
    """SVGDOptimizer.step() with REAL gradients: every particle's backward produces n_tensors fresh gradient tensors
    (one pre-built autograd node, _ManyGrads).  ONE loop of whole steps is timed; inside it the closures carry their
    own host timers, so the optimizer's host time = step host time - closure host time of the SAME steps (never
    negative), and HIP events around the posterior update give its GPU time.  No gradient is copied: the kernels read
    the tensors autograd produced (svgd.py:129-133's clones removed).

    ctor = "opt_in": round 3's figure -- equal-sized tensors, nesterov SGD, fuse_base_optimizer=True, reuse_gram=True.
    ctor = "reference": the optimizer built EXACTLY as the reference builds it (experiments/iwildcam/models.py:120 with
    iwildcam.yaml:214-221): ``SVGDOptimizer(model.parameters(), reset_model, Adam(model.parameters(), lr=3e-5,
    weight_decay=0), particle_count=..., l2_reg=0.0, dataset_size=129809, kernel_grad_scale=1.0)`` -- no extra keyword --
    over the 161 tensors of the iWildCam ResNet-50 in their real shapes; reset_model re-initialises the head only.
    ctor = "reference_unfused": the same with fuse_base_optimizer=False (particle_count x torch Adam.step per step, the
    reference's own cost structure, svgd.py:99-103)."""
    import beyond_deep_ensembles_amd as bde
    if ctor == "opt_in":
        sizes = [d // n_tensors] * (n_tensors - 1)
        sizes.append(d - sum(sizes))
        shapes = [(n,) for n in sizes]
    else:
        shapes = resnet50_shapes()
        n_tensors, d = len(shapes), sum(int(torch.Size(sh).numel()) for sh in shapes)
    params = [torch.nn.Parameter(torch.randn(sh, device=dev) * 0.05) for sh in shapes]
    cs = [torch.randn(sh, device=dev) * 0.01 for sh in shapes]

    def reset():
        with torch.no_grad():
            for p in params[-2:]:
                p.normal_(0, 0.05)
    if ctor == "opt_in":
        base = torch.optim.SGD(params, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=3e-4)
        opt = bde.SVGDOptimizer(params, reset, base, particle_count=particles, dataset_size=DATASET_SIZE,
                                fuse_base_optimizer=fuse, reuse_gram=fuse)
    else:
        svgd_cfg = dict(particle_count=particles, l2_reg=0.0, dataset_size=129809, kernel_grad_scale=1.0)   # iwildcam.yaml:217-221
        base_cfg = dict(lr=0.00003, weight_decay=0)                                                          # iwildcam.yaml:214-216
        extra_kw = {"fuse_base_optimizer": False} if ctor == "reference_unfused" else {}
        opt = bde.SVGDOptimizer(iter(params), reset, torch.optim.Adam(iter(params), **base_cfg), **svgd_cfg, **extra_kw)
        fuse = bool(opt._fuse)
    t_closures = [0.0]

    def fwd():
        t0 = time.perf_counter()
        out = _ManyGrads.apply(cs, *params)
        t_closures[0] += time.perf_counter() - t0
        return out

    def bwd(loss):
        t0 = time.perf_counter()
        loss.backward()
        t_closures[0] += time.perf_counter() - t0
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    inner = opt._posterior_update
    it = [0]

    def timed_update(*a, **k):
        i = it[0]
        if 0 <= i < steps:
            ev[i][0].record()
        out = inner(*a, **k)
        if 0 <= i < steps:
            ev[i][1].record()
        it[0] += 1
        return out
    opt._posterior_update = timed_update
    it[0] = -3
    for _ in range(3):
        opt.step(fwd, bwd)
    torch.cuda.synchronize()
    t_closures[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.step(fwd, bwd)
    host = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    clos = t_closures[0] / steps
    gpu = sum(a.elapsed_time(b) for a, b in ev) / steps
    del opt, params, cs
    torch.cuda.empty_cache()
    return {"step_ms": round(wall * 1e3, 3), "step_host_ms": round(host * 1e3, 3), "closures_host_ms": round(clos * 1e3, 3),
            "optimizer_host_ms": round((host - clos) * 1e3, 3), "optimizer_gpu_ms": round(gpu, 3),
            "tensors": n_tensors, "particles": particles, "fused": fuse, "constructor": ctor,
            "what": "SVGDOptimizer.step with real gradients (8 backward passes producing n_tensors fresh tensors each), one "
                    "timed loop: host time of the whole step, of the closures inside it, their difference = the "
                    "optimizer's own host time (re-pointing, gradient hand-over by reference, launches), and the GPU time "
                    "of the posterior update (HIP events); no per-particle gradient copy"}


def other_shell_steps_ms(dev, steps=10):
    """iVONOptimizer.step and SwagOptimizer.step over 161 parameter tensors totalling ResNet-50 size; the forward
    closure returns a constant and the backward closure hands out preallocated gradient tensors (no model work): what
    is timed is the optimizer's own work per training step -- host logic + kernels (+ torch's SGD for SWAG)."""
    import beyond_deep_ensembles_amd as bde
    n_tensors, d = 161, D_RESNET50
    sizes = [d // n_tensors] * (n_tensors - 1)
    sizes.append(d - sum(sizes))
    out = {}
    zero = torch.zeros((), device=dev)

    def make():
        params = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.05) for s in sizes]
        grads = [torch.randn(s, device=dev) * 0.01 for s in sizes]

        def bwd(loss):
            for p, g in zip(params, grads):
                p.grad = g
        return params, bwd

    def run(opt, bwd):
        for _ in range(3):
            opt.step(lambda: zero, bwd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            opt.step(lambda: zero, bwd)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    params, bwd = make()
    opt = bde.iVONOptimizer(params, lr=1e-3, prior_prec=100.0, dataset_size=DATASET_SIZE, mc_samples=2, rng="philox")
    out["ivon_step_mc2"] = {"ms": round(run(opt, bwd) * 1e3, 3),
                            "what": "iVONOptimizer.step, 2 MC samples: 2 weight-noise draws + gradient hand-over + update; "
                                    "kernels alone 2 x 0.073 + 0.116 ms"}
    del opt, params
    params, bwd = make()
    opt = bde.SwagOptimizer(params, torch.optim.SGD(params, lr=1e-3, momentum=0.9), update_interval=1, deviation_samples=K_SWAG,
                            rng="philox")
    out["swag_step_update_every_step"] = {"ms": round(run(opt, bwd) * 1e3, 3),
                                          "what": "SwagOptimizer.step with a moment update on EVERY step (the reference updates "
                                                  "every 255...3300 steps): torch SGD.step over 161 tensors + bde_swag_update "
                                                  "(0.088 ms)"}
    del opt, params
    out["tensors"] = n_tensors
    return out


def config_extras(dev, on_section=None):
    """The other BASELINE.json configs through the PRODUCT shells (null closures: what is timed is the optimizer's
    own work -- kernels + host logic -- per step / per posterior sample):
      configs[1]  CIFAR ResNet-20 SVGD, 8 particles: SVGDOptimizer.step, SGD-nesterov base (cifar.yaml), unfused and fused;
      configs[2]  CIFAR ResNet-20 SWAG, K = 20, 30 posterior samples: DeepEnsemble.predict through the batched sampler;
      configs[4]  Camelyon DenseNet-121 MultiSWAG, 5 modes x 30 samples: DeepEnsemble.predict, this GPU's share at N = 1."""
    import beyond_deep_ensembles_amd as bde
    out = {}
    zero = torch.zeros((), device=dev)

    def tensors(d, n):
        sizes = [d // n] * (n - 1)
        sizes.append(d - sum(sizes))
        return [torch.nn.Parameter(torch.randn(s, device=dev) * 0.05) for s in sizes]

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    # ---- configs[1]: 65 tensors like the swish/FRN ResNet-20
    for fuse in (False, True):
        params = tensors(D_RESNET20, 65)
        base = torch.optim.SGD(params, lr=0.1, momentum=0.9, nesterov=True, weight_decay=3e-4)

        def reset():
            with torch.no_grad():
                params[-1].normal_(0, 0.05)
        opt = bde.SVGDOptimizer(params, reset, base, particle_count=M, dataset_size=50000.0, l2_reg=3e-4,
                                fuse_base_optimizer=fuse)
        t = timed(lambda: opt.step(lambda: zero, lambda loss: None), 30)
        out["svgd_step_cifar_resnet20_shell_" + ("fused" if fuse else "unfused")] = {
            "ms": round(t * 1e3, 4), "steps_per_s": round(1.0 / t, 1), "tensors": 65, "particles": M,
            "small_model_kernel": bool(opt._small_model(M, opt._layout.d)),
            "what": "SVGDOptimizer.step as a default constructor builds it, null closures: the streaming update (Gram -> "
                    "statistics -> " + ("fused update incl. 8 SGD applications)" if fuse else "combine) + 8 x torch SGD.step") +
                    "; the small-model kernel takes over once device_verified.json records it: see ..._small_kernel"}
        del opt, params, base

    # ---- the BBB layer forward of the iWildCam head (2048 -> 182) at batch 16: fused op vs the reference's op sequence
    import torch.nn.functional as F
    prior = bde.GaussianPrior(0, 1.0)
    for name, (bsz, fi, fo) in {"iwildcam_head_b16": (16, 2048, 182), "uci_mlp_layer_b32": (32, 13, 50),
                                "mlp_4096x4096_b64": (64, 4096, 4096)}.items():
        layer = bde.BBBLinear(fi, fo, prior, prior, rng="philox").to(dev)
        xin = torch.randn(bsz, fi, device=dev)

        def torch_sequence():                                   # bbb_layers.py:70-80 as ATen ops
            w, b = layer.weight, layer.bias
            mean = F.linear(xin, w.mean, b.mean)
            var = F.linear((xin ** 2).clamp(min=1e-4), (w.std ** 2).clamp(min=1e-4), (b.std ** 2).clamp(min=1e-4))
            return mean + torch.sqrt(var) * torch.empty_like(mean).normal_(0, 1)
        with torch.no_grad():
            t_f = time_loop(lambda: layer(xin), 50)
            t_t = time_loop(torch_sequence, 50)
        out["bbb_linear_forward_" + name] = {"ms": round(t_f * 1e3, 4), "torch_sequence_ms": round(t_t * 1e3, 4),
                                             "speedup": round(t_t / t_f, 2), "B": bsz, "I": fi, "O": fo,
                                             "weights_GBps": round(8.0 * fi * fo / t_f / 1e9, 1),
                                             "what": "BBBLinear.forward (training mode, in-kernel noise): bde_lrt_linear_fwd "
                                                     "vs the reference's ~14 ATen launches"}
        # forward + backward (weights and input): bde_lrt_linear_fwd + bde_lrt_linear_bwd vs autograd over the sequence
        xg = xin.clone().requires_grad_(True)
        leaves = [xg, layer.weight.mean, layer.weight.rho, layer.bias.mean, layer.bias.rho]

        def fused_fb():
            torch.autograd.grad(layer(xg).sum(), leaves)

        def torch_fb():
            w, b = layer.weight, layer.bias
            mean = F.linear(xg, w.mean, b.mean)
            var = F.linear((xg ** 2).clamp(min=1e-4), (w.std ** 2).clamp(min=1e-4), (b.std ** 2).clamp(min=1e-4))
            torch.autograd.grad((mean + torch.sqrt(var) * torch.empty_like(mean).normal_(0, 1)).sum(), leaves)
        t_f = time_loop(fused_fb, 30)
        t_t = time_loop(torch_fb, 30)
        out["bbb_linear_fwd_bwd_" + name] = {"ms": round(t_f * 1e3, 4), "torch_sequence_ms": round(t_t * 1e3, 4),
                                             "speedup": round(t_t / t_f, 2), "B": bsz, "I": fi, "O": fo,
                                             "weights_GBps": round(28.0 * fi * fo / t_f / 1e9, 1),
                                             "what": "BBBLinear forward + backward (all five gradients): 2 + 3..4 launches, "
                                                     "28*O*I algorithmic bytes, vs autograd over the reference's op sequence"}
        if name == "mlp_4096x4096_b64":
            # the C-ABI ops alone (kernels, preallocated outputs), with the sigma^2 cache of the weight version
            ops = layer._ops if hasattr(layer, "_ops") else None
            from beyond_deep_ensembles_amd.ops import HipOps
            ops = ops or HipOps()
            wm, wr, bm, br = (t.detach() for t in (layer.weight.mean, layer.weight.rho, layer.bias.mean, layer.bias.rho))
            o_, v_ = torch.empty(bsz, fo, device=dev), torch.empty(bsz, fo, device=dev)
            gg = torch.randn(bsz, fo, device=dev)
            gx_, gwm_, gwr_ = torch.empty_like(xin), torch.empty_like(wm), torch.empty_like(wr)
            gbm_, gbr_ = torch.empty_like(bm), torch.empty_like(br)
            s2_, ds2_ = torch.empty_like(wr), torch.empty_like(wr)
            t_c = time_loop(lambda: ops.lrt_sigma_cache(wr, s2_, ds2_), 30)
            t_kf = time_loop(lambda: ops.lrt_linear_fwd(xin, wm, wr, bm, br, True, o_, v_, seed=1, stream_id=2, w_s2=s2_), 50)
            t_kb = time_loop(lambda: ops.lrt_linear_bwd(xin, wm, wr, br, True, gg, v_, gx_, gwm_, gwr_, gbm_, gbr_, seed=1,
                                                        stream_id=2, w_s2=s2_, w_ds2=ds2_), 50)
            out["bbb_linear_kernels_" + name] = {
                "ms": round((t_kf + t_kb) * 1e3, 4), "fwd_us": round(t_kf * 1e6, 1), "bwd_us": round(t_kb * 1e6, 1),
                "sigma_cache_us": round(t_c * 1e6, 1), "B": bsz, "I": fi, "O": fo,
                "what": "bde_lrt_linear_fwd / bde_lrt_linear_bwd alone (2 / 4 launches, preallocated outputs) with the "
                        "sigma^2 cache of the weight version (bde_lrt_sigma_cache: once per base_optimizer.step)"}
            del o_, v_, gg, gx_, gwm_, gwr_, s2_, ds2_
        del layer

    # ---- BBBLinear ABOVE the fused op's 128 rows per launch: ceil(B / 128) launches of the same (device-verified) kernels
    # (fused_linear_max_rows) vs what a default-constructed layer runs there (two stock GEMMs + fused element-wise passes) vs the
    # reference's op sequence -- the measurement that decides whether the default limit moves (BBBLinear)
    for bsz in (256, 512, 1024):
        key = f"bbb_linear_fwd_bwd_mlp_4096x4096_b{bsz}_row_tiles"
        try:
            tiled = bde.BBBLinear(4096, 4096, prior, prior, rng="philox", fused_linear_max_rows=1024).to(dev)
            stock = bde.BBBLinear(4096, 4096, prior, prior, rng="philox").to(dev)
            stock.load_state_dict(tiled.state_dict())
            xg = torch.randn(bsz, 4096, device=dev, requires_grad=True)

            def fb(layer):
                torch.autograd.grad(layer(xg).sum(), [xg, layer.weight.mean, layer.weight.rho, layer.bias.mean, layer.bias.rho])

            def torch_fb():
                w, b = stock.weight, stock.bias
                mean = F.linear(xg, w.mean, b.mean)
                var = F.linear((xg ** 2).clamp(min=1e-4), (w.std ** 2).clamp(min=1e-4), (b.std ** 2).clamp(min=1e-4))
                torch.autograd.grad((mean + torch.sqrt(var) * torch.empty_like(mean).normal_(0, 1)).sum(),
                                    [xg, w.mean, w.rho, b.mean, b.rho])
            t_tiled, t_stock, t_torch = time_loop(lambda: fb(tiled), 20), time_loop(lambda: fb(stock), 20), time_loop(torch_fb, 20)
            out[key] = {"ms": round(t_tiled * 1e3, 4), "default_layer_ms": round(t_stock * 1e3, 4),
                        "torch_sequence_ms": round(t_torch * 1e3, 4), "speedup_vs_default_layer": round(t_stock / t_tiled, 2),
                        "speedup_vs_torch_sequence": round(t_torch / t_tiled, 2), "B": bsz, "I": 4096, "O": 4096,
                        "tiles": (bsz + 127) // 128, "TFLOPs": round(12.0 * bsz * 4096 * 4096 / t_tiled / 1e12, 1),
                        "what": "BBBLinear(fused_linear_max_rows=1024) forward + backward in row tiles of 128 (bde_lrt_linear_fwd / "
                                "_bwd per tile, weight gradients accumulated by autograd) vs the default layer at this batch (stock "
                                "GEMMs + fused element-wise passes) vs the reference's op sequence"}
            del tiled, stock, xg
        except Exception as e:                                       # noqa: BLE001
            out[key] = {"error": f"{type(e).__name__}: {e}"}

    # ---- configs[0]: BBBOptimizer.step on the UCI-housing MLP (13 -> 50 -> 1 BBBLinear, 5 MC samples, Adam), whole
    # step incl. forward/backward; beside it the reference's op sequence for the same step in plain PyTorch on this GPU
    def uci_model():
        return torch.nn.Sequential(bde.BBBLinear(13, 50, prior, prior, rng="philox"), torch.nn.ReLU(),
                                   bde.BBBLinear(50, 1, prior, prior, rng="philox")).to(dev)
    xb, yb = torch.randn(32, 13, device=dev), torch.randn(32, 1, device=dev)
    model = uci_model()
    opt = bde.BBBOptimizer(model.parameters(), torch.optim.Adam(model.parameters(), lr=1e-3), prior, dataset_size=455,
                           mc_samples=5)
    t_ours = time_loop(lambda: opt.step(lambda: F.mse_loss(model(xb), yb), lambda loss: loss.backward()), 30)
    ref = uci_model()
    ref_layers = [ref[0], ref[2]]
    ref_adam = torch.optim.Adam(ref.parameters(), lr=1e-3)

    def ref_forward():                                           # bbb_layers.py:62-87 (the CUDA branch, incl. the per-forward KL)
        hcur = xb
        for li, layer in enumerate(ref_layers):
            w, b = layer.weight, layer.bias
            w_std, b_std = F.softplus(w.rho), F.softplus(b.rho)
            batch_in = torch.stack((hcur, (hcur ** 2).clamp(min=1e-4)))
            batch_mat = torch.stack((w.mean.transpose(0, 1), (w_std.transpose(0, 1) ** 2).clamp(min=1e-4)))
            batch_add = torch.stack((b.mean.expand((hcur.shape[0], w.mean.shape[0])),
                                     (b_std ** 2).clamp(min=1e-4).expand((hcur.shape[0], w.mean.shape[0]))))
            both = torch.baddbmm(batch_add, batch_in, batch_mat)
            hcur = both[0] + torch.sqrt(both[1]) * torch.empty_like(both[0]).normal_(0, 1)
            layer._ref_kl = prior.kl_divergence(w.mean, w_std) + prior.kl_divergence(b.mean, b_std)
            if li == 0:
                hcur = F.relu(hcur)
        return F.mse_loss(hcur, yb)

    def ref_step():                                              # bbb.py:59-89
        ref_adam.zero_grad()
        total = None
        for _ in range(5):
            total = ref_forward() if total is None else total + ref_forward()
        kl = torch.tensor(0.0, device=dev)
        for layer in ref_layers:
            for gp in (layer.weight, layer.bias):
                kl += prior.kl_divergence(gp.mean, F.softplus(gp.rho))
        loss = (1.0 / 455) * kl + total / 5
        if not loss.isnan().any():
            loss.backward()
            ref_adam.step()
        return loss
    t_ref = time_loop(ref_step, 30)
    out["bbb_step_uci_mlp_mc5"] = {"ms": round(t_ours * 1e3, 4), "torch_sequence_ms": round(t_ref * 1e3, 4),
                                   "speedup": round(t_ref / t_ours, 2), "steps_per_s": round(1.0 / t_ours, 1),
                                   "what": "BASELINE configs[0]: BBBOptimizer.step (5 MC samples, batch 32, Adam) on the UCI "
                                           "MLP of BBBLinear layers, forward and backward included: fused layer forward / "
                                           "backward ops + one KL launch, vs the reference's op sequence (bbb_layers.py:62-87, "
                                           "bbb.py:59-89) in PyTorch on this GPU; both are host-bound"}
    del model, opt, ref, ref_adam

    # ---- SWAG members with all K columns filled
    def swag_member(d, n_tensors, seed):
        params = tensors(d, n_tensors)
        opt = bde.SwagOptimizer(params, torch.optim.SGD(params, lr=1e-3), update_interval=1, deviation_samples=K_SWAG,
                                rng="philox", seed=seed)
        with torch.no_grad():
            for _ in range(K_SWAG + 2):
                opt._theta[:d] += torch.randn(d, device=dev) * 1e-3
                opt._swag_update()
        model = torch.nn.Module()
        model.p = torch.nn.ParameterList(params)
        return model, opt

    ens = bde.DeepEnsemble([swag_member(D_RESNET20, 65, 1)])
    t = timed(lambda: ens.predict(lambda m: zero, S_SWAG), 20)
    out["swag_predict_30_samples_cifar_resnet20"] = {"ms": round(t * 1e3, 4), "samples_per_s": round(S_SWAG / t, 1),
                                                     "K": K_SWAG, "what": "DeepEnsemble.predict(30): one batched MFMA "
                                                     "sampling pass, each sample served by one device copy into the vector "
                                                     "the 65 parameters view; null predict closure"}
    del ens
    ens = bde.DeepEnsemble([swag_member(D_DENSENET, 364, 10 + i) for i in range(5)])
    t = timed(lambda: ens.predict(lambda m: zero, 5 * S_SWAG), 5)
    out["multiswag_predict_5x30_camelyon_densenet121"] = {
        "ms": round(t * 1e3, 3), "samples_per_s": round(5 * S_SWAG / t, 1), "members": 5, "K": K_SWAG, "D": D_DENSENET,
        "what": "DeepEnsemble.predict(150) over 5 SWAG members (364 tensors each): 5 batched sampling passes, each sample "
                "served by one device copy; null predict closure; with N GPUs each rank does 1/N of the units "
                "(predict_distributed)"}
    del ens
    torch.cuda.empty_cache()
    if on_section is not None:
        on_section(out)                 # everything above has been measured; LAST: kernels that have never run on an MI355X
    # ---- configs[1] again through the small-model kernel + the round-5 native host paths (what the default becomes once they are
    # device-verified), then with graph_replay=True on top: never run on an MI355X -- each entry fails on its own
    small = dict(single_launch="two", host_fast_paths=True)
    for fuse in (True, False):
        key = "svgd_step_cifar_resnet20_shell_" + ("fused" if fuse else "unfused") + "_small_kernel"
        try:
            params = tensors(D_RESNET20, 65)
            base = torch.optim.SGD(params, lr=0.1, momentum=0.9, nesterov=True, weight_decay=3e-4)

            def reset_small():
                with torch.no_grad():
                    params[-1].normal_(0, 0.05)
            opt = bde.SVGDOptimizer(params, reset_small, base, particle_count=M, dataset_size=50000.0, l2_reg=3e-4,
                                    fuse_base_optimizer=fuse, **small)
            t = timed(lambda: opt.step(lambda: zero, lambda loss: None), 30)
            out[key] = {"ms": round(t * 1e3, 4), "steps_per_s": round(1.0 / t, 1), "tensors": 65, "particles": M,
                        "small_model_kernel": bool(opt._small_model(M, opt._layout.d)),
                        "what": "as svgd_step_cifar_resnet20_shell_" + ("fused" if fuse else "unfused") + " with "
                                "single_launch='two', host_fast_paths=True: gradient packing + the small-model kernel's two launches"}
            del opt, params, base
        except Exception as e:                                       # noqa: BLE001 -- reported, the other sections go on
            out[key] = {"error": f"{type(e).__name__}: {e}"}
    if on_section is not None:
        on_section(out)
    try:
        params = tensors(D_RESNET20, 65)
        base = torch.optim.SGD(params, lr=0.1, momentum=0.9, nesterov=True, weight_decay=3e-4)

        def reset_last():
            with torch.no_grad():
                params[-1].normal_(0, 0.05)
        opt = bde.SVGDOptimizer(params, reset_last, base, particle_count=M, dataset_size=50000.0, l2_reg=3e-4, graph_replay=True,
                                **small)
        for _ in range(8):                                           # the eager first steps + one recording per staging slot
            opt.step(lambda: zero, lambda loss: None)
        t = timed(lambda: opt.step(lambda: zero, lambda loss: None), 30)
        out["svgd_step_cifar_resnet20_shell_fused_graph_replay"] = {
            "ms": round(t * 1e3, 4), "steps_per_s": round(1.0 / t, 1), "replayed_steps": opt._graph_replays,
            "recordings": opt._graph_captures,
            "what": "as svgd_step_cifar_resnet20_shell_fused_small_kernel with graph_replay=True: table upload + gradient "
                    "packing + the update's two launches as ONE hipGraph launch per step"}
        del opt, params, base
    except Exception as e:                                           # noqa: BLE001 -- reported, the other sections go on
        out["svgd_step_cifar_resnet20_shell_fused_graph_replay"] = {"error": f"{type(e).__name__}: {e}"}
    if on_section is not None:
        on_section(out)
    # ---- the BBBConv2d layers of the CIFAR ResNet-20 (BASELINE configs[1] model family, batch 128): the fused layer
    # (bde_conv_lrt_fwd + bde_conv_lrt_bwd_data / _weight: each pair of convolutions of bbb_layers.py:146-147 as ONE
    # dual-accumulator implicit GEMM) vs the reference's op sequence (two MIOpen convolutions + element-wise ops) and vs
    # round 3's composition (stock convolutions, fused element-wise passes: fused_conv=False)
    from beyond_deep_ensembles_amd import bbb_layers as _bl
    conv_shapes = {"resnet20_layer_b128": (128, 16, 32, 32, 16, 3, 1, 1), "resnet20_first_b128": (128, 3, 32, 32, 16, 3, 1, 1),
                   "resnet20_16to32_s2_b128": (128, 16, 32, 32, 32, 3, 2, 1), "resnet20_32ch_b128": (128, 32, 16, 16, 32, 3, 1, 1),
                   "resnet20_32to64_s2_b128": (128, 32, 16, 16, 64, 3, 2, 1), "resnet20_64ch_b128": (128, 64, 8, 8, 64, 3, 1, 1)}
    for cname, shape in conv_shapes.items():
        # one geometry failing (these kernels have never run on an MI355X) must not cost the geometries behind it
        try:
            out["bbb_conv2d_fwd_bwd_" + cname] = conv_layer_entry(dev, prior, shape)
        except Exception as e:                                       # noqa: BLE001
            out["bbb_conv2d_fwd_bwd_" + cname] = {"error": f"{type(e).__name__}: {e}"}
        if on_section is not None:
            on_section(out)

    return out


def conv_layer_entry(dev, prior, shape):
    """One BBBConv2d geometry: the device-verified compositions FIRST (the reference's op sequence, round 3's stock convolutions
    + fused element-wise passes = what a default-constructed layer runs), then the fused convolution kernels forced on -- a
    failure there is recorded beside the stock numbers instead of replacing them."""
    import torch.nn.functional as F
    import beyond_deep_ensembles_amd as bde
    from beyond_deep_ensembles_amd import bbb_layers as _bl
    cn, cc_, chh, cww, co, ck, cs, cp = shape
    conv = bde.BBBConv2d(cc_, co, ck, prior, prior, stride=cs, padding=cp, rng="philox", fused_conv=True).to(dev)
    conv3 = bde.BBBConv2d(cc_, co, ck, prior, prior, stride=cs, padding=cp, rng="philox", fused_conv=False).to(dev)
    conv3.load_state_dict(conv.state_dict())
    xc = torch.randn(cn, cc_, chh, cww, device=dev, requires_grad=True)

    def leaves(layer):
        return [xc, layer.weight.mean, layer.weight.rho, layer.bias.mean, layer.bias.rho]

    def conv_fused():
        torch.autograd.grad(conv(xc).sum(), leaves(conv))

    def conv_round3():
        torch.autograd.grad(conv3(xc).sum(), leaves(conv3))

    def conv_torch():                                            # bbb_layers.py:146-154
        w, b = conv.weight, conv.bias
        mean = F.conv2d(xc, w.mean, b.mean, stride=cs, padding=cp)
        var = F.conv2d((xc ** 2).clamp(min=1e-4), (w.std ** 2).clamp(min=1e-4), b.std ** 2, stride=cs, padding=cp)
        torch.autograd.grad((mean + torch.sqrt(var) * torch.empty_like(mean).normal_(0, 1)).sum(), leaves(conv))
    t_3, t_t = time_loop(conv_round3, 30), time_loop(conv_torch, 30)
    ho, wo = (chh + 2 * cp - ck) // cs + 1, (cww + 2 * cp - ck) // cs + 1
    flops_fwd = 2 * 2.0 * cn * co * ho * wo * cc_ * ck * ck
    entry = {
        "ms": round(t_3 * 1e3, 4), "torch_sequence_ms": round(t_t * 1e3, 4), "speedup": round(t_t / t_3, 2),
        "round3_composition_ms": round(t_3 * 1e3, 4),
        "shape": {"N": cn, "C": cc_, "H": chh, "W": cww, "O": co, "K": ck, "stride": cs, "padding": cp},
        "what": "BBBConv2d forward + backward (all five gradients) through the layer.  ms / speedup: what a default-constructed "
                "layer runs at this geometry (see default_path) vs the reference's op sequence under autograd (two MIOpen "
                "convolutions forward, four backward, ~25 element-wise launches).  round3_composition_ms: stock convolutions + "
                "fused element-wise passes.  fused_kernels: the dual-accumulator implicit-GEMM kernels forced on (1 forward "
                "launch; g_var + input-gradient + weight-gradient + finish launches backward)",
        "native_autograd_nodes": _bl._native_nodes(conv.weight._get_ops()) is not None,
        # what BBBConv2d() without a keyword does at this geometry: the fused kernels only where conv_profit.json records
        # a device measurement of this kernel version that beats the stock sequence (forward + backward)
        "default_path": "fused" if _bl._conv_profitable((cn, cc_, chh, cww), (co, cc_, ck, ck), (cs, cs), (cp, cp),
                                                         conv.weight._get_ops(), True) else "stock (round-3 composition)"}
    try:
        with torch.no_grad():
            t_ff = time_loop(lambda: conv(xc), 30)
        t_f = time_loop(conv_fused, 30)
        entry["fused_kernels"] = {"ms": round(t_f * 1e3, 4), "speedup_vs_torch_sequence": round(t_t / t_f, 2),
                                  "speedup_vs_round3_composition": round(t_3 / t_f, 2), "forward_only_ms": round(t_ff * 1e3, 4),
                                  "forward_TFLOPs": round(flops_fwd / t_ff / 1e12, 2)}
        if entry["default_path"] == "fused":
            entry["ms"], entry["speedup"] = round(t_f * 1e3, 4), round(t_t / t_f, 2)
    except Exception as e:                                           # noqa: BLE001
        entry["fused_kernels"] = {"error": f"{type(e).__name__}: {e}"}
    return entry



def single_gpu_extras(ops, dev, args, sink=None):
    """Everything under `extra` that one process on one GPU measures (every other kernel at ResNet-50 size, the product
    shells' steps, the other BASELINE configs); each section catches its own exceptions.  ``sink``: a file that receives
    the sections finished so far after each of them (the child process of extras_in_child: what was measured before a
    fault survives it)."""
    ex = {}

    def checkpoint():
        if sink:
            with open(sink, "w") as fh:
                json.dump(ex, fh)

    def first_sections(part):
        ex.update(part)
        checkpoint()
    try:
        ex.update(extras(ops, dev, quick=False, on_section=first_sections))
    except Exception as e:                  # the headline line above must survive a failing extra
        import traceback
        ex.update({"error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()[-1500:]})
        log(f"  extras failed: {ex['error']}")
    checkpoint()
    try:
        ex["svgd_shell_step_ms"] = shell_step_ms(dev)
        log(f"  svgd_shell_step_ms {ex['svgd_shell_step_ms']}")
        if not args.no_config_extras:       # other sizes: kept out of the PMC passes (one size per kernel there)
            ex["svgd_shell_step_densenet121_ms"] = shell_step_ms(dev, n_tensors=364, d=D_DENSENET)
            log(f"  svgd_shell_step_densenet121_ms {ex['svgd_shell_step_densenet121_ms']}")
    except Exception as e:
        log(f"  svgd_shell_step_ms skipped: {e}")
    checkpoint()
    try:
        ex["svgd_shell_step_real_grads_ms"] = shell_step_real_grads_ms(dev)
        if not args.no_config_extras:
            ex["svgd_shell_step_real_grads_densenet121_ms"] = \
                shell_step_real_grads_ms(dev, 364, D_DENSENET)
        log(f"  svgd_shell_step_real_grads_ms {ex['svgd_shell_step_real_grads_ms']}")
        # the drop-in a user of the reference actually gets: the reference's own constructor call
        ref_ctor = {"m8": shell_step_real_grads_ms(dev, ctor="reference"),
                    "m8_fuse_base_optimizer_false": shell_step_real_grads_ms(dev, ctor="reference_unfused", steps=6),
                    "m5_reference_particle_count": shell_step_real_grads_ms(dev, ctor="reference", particles=5)}
        base_step = ex["svgd_shell_step_real_grads_ms"]["step_ms"]
        ref_ctor["m8_vs_opt_in_step"] = round(ref_ctor["m8"]["step_ms"] / base_step, 3)
        ref_ctor["what"] = ("SVGDOptimizer built exactly as experiments/iwildcam/models.py:120 builds it (Adam lr 3e-5, "
                            "no extra keyword -> fuse_base_optimizer='auto'), real ResNet-50 tensor shapes, real "
                            "gradients; m8_vs_opt_in_step = its step time / svgd_shell_step_real_grads_ms.step_ms "
                            "(fused nesterov SGD + reuse_gram, opt-in keywords)")
        ex["svgd_reference_constructor_step"] = ref_ctor
        log(f"  svgd_reference_constructor_step {ref_ctor}")
    except Exception as e:
        log(f"  svgd_shell_step_real_grads_ms skipped: {type(e).__name__}: {e}")
    checkpoint()
    try:
        ex["other_shell_steps_ms"] = other_shell_steps_ms(dev)
        log(f"  other_shell_steps_ms {ex['other_shell_steps_ms']}")
    except Exception as e:
        log(f"  other_shell_steps_ms skipped: {type(e).__name__}: {e}")
    checkpoint()
    try:
        # LAST: this section launches kernels that have never run on an MI355X (the fused BBBConv2d kernels, forced on)
        if not args.no_config_extras:
            def config_sections(part):
                ex["other_baseline_configs"] = dict(part)
                checkpoint()
            ex["other_baseline_configs"] = config_extras(dev, on_section=config_sections)
            for k, v in ex["other_baseline_configs"].items():
                log(f"  {k}: {v.get('ms', v.get('error'))} ms")
    except Exception as e:
        log(f"  other_baseline_configs skipped: {type(e).__name__}: {e}")
    checkpoint()
    return ex


def extras_in_child(args, dev_index, limit_s=600):
    """`extra` measured by a CHILD process (`python bench.py --extras-child FILE`): a Python exception in an extra is caught
    section by section, but a GPU fault or a hang in one of them -- several run kernels that have never executed on an
    MI355X -- would take the process down before the headline line is printed.  The parent has
    finished (and freed) the headline workload when this starts; the child is a new process (no exec of this one), gets
    `limit_s` seconds, and its exact PID is killed on a timeout."""
    import subprocess
    import tempfile
    fd, path = tempfile.mkstemp(suffix=".json", prefix="bde_bench_extra_")
    os.close(fd)
    cmd = [sys.executable, os.path.abspath(__file__), "--extras-child", path, "--gpus", "1"]
    if args.no_config_extras:
        cmd.append("--no-config-extras")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "GROUP_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env["BDE_BENCH_DEVICE"] = str(dev_index)
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL)       # its log lines go to our stderr
        timed_out = False
        try:
            rc = proc.wait(timeout=limit_s)
        except subprocess.TimeoutExpired:
            timed_out = True
            proc.kill()
            rc = proc.wait()
        try:
            with open(path) as f:                                              # (the child rewrites it after every section)
                out = json.load(f)
        except (OSError, ValueError):
            out = {}
        if timed_out:
            out["error"] = (f"the extras child did not finish within {limit_s} s and was killed" +
                            (" after writing these sections" if out else ""))
        elif rc != 0:
            out["error"] = f"the extras child exited with code {rc}" + (" after writing these sections" if out else "")
        return out
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


def being_profiled() -> bool:
    """Is this process running under rocprofv3 (its tool library preloaded)?  Such a process has initialised the GPU before
    main() starts and must not create further processes."""
    return "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or bool(os.environ.get("ROCP_TOOL_LIBRARIES"))


def parse_pmc_counter(paths, counter, kernel_substring):
    """Mean of `counter` over the dispatches of kernels whose name contains `kernel_substring` in rocprofv3
    `*counter_collection.csv` files -> (mean, dispatches)."""
    import csv
    vals = []
    for path in paths:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter and kernel_substring in row.get("Kernel_Name", ""):
                    vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def pmc_traffic_bytes(fetch_kib, write_kib):
    """HBM bytes per launch from the two counters as MI355X_MICROARCH.md prescribes: both are in KiB; on gfx950 FETCH_SIZE
    tallies exactly half the bytes of a wide coalesced streaming read (16 B per lane), so it is doubled."""
    return int(round(2.0 * fetch_kib * 1024.0 + write_kib * 1024.0))


def live_traffic(d, dev_index, limit_s=120):
    """roofline.traffic measured IN THIS RUN (it used to be a constant recorded once per round): two CHILD
    passes, `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separately: the two do not fit one pass), each over
    `python3 bench.py --traffic-child` = a few full-size steps of the headline's kernels; the combine kernel's mean counter
    values -> bytes per launch.  Counters only (no trace flags), the program itself behind `--`, run from /tmp with TMPDIR=/tmp,
    as the profiling guide prescribes.  Returns (bytes, source) or (None, reason): any failure leaves the recorded value in place."""
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    if being_profiled():
        return None, "this process is itself being profiled"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "GROUP_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env["TMPDIR"] = "/tmp"
    env["BDE_BENCH_DEVICE"] = str(dev_index)
    means = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix="bde_pmc_", dir="/tmp")
        cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "p", "--",
               sys.executable, os.path.abspath(__file__), "--traffic-child", "--dim", str(d)]
        try:
            proc = subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            try:
                rc = proc.wait(timeout=limit_s)
            except subprocess.TimeoutExpired:
                proc.kill()
                proc.wait()
                return None, f"the {counter} pass did not finish within {limit_s} s"
            if rc != 0:
                return None, f"the {counter} pass exited with code {rc}"
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            mean, n = parse_pmc_counter(files, counter, "svgd_combine_kernel")
            if mean is None:
                return None, f"no {counter} rows for the combine kernel"
            means[counter] = (mean, n)
        except OSError as e:
            return None, f"{type(e).__name__}: {e}"
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    nbytes = pmc_traffic_bytes(means["FETCH_SIZE"][0], means["WRITE_SIZE"][0])
    return nbytes, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, two separate child passes over "
                    f"{means['FETCH_SIZE'][1]} / {means['WRITE_SIZE'][1]} launches of the kernel at this size (bench.py --traffic-child); "
                    "KiB counters, FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md")


def timed_blocks(step, steps, blocks, dist, dev):
    """`blocks` blocks of exactly `steps` steps, each bracketed by barrier + synchronize; MAX over ranks per block."""
    out = []
    for _ in range(blocks):
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist:
            tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        out.append(elapsed / steps * 1e3)
    return out


def median(v):
    s = sorted(v)
    return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


class _NoKernels:
    """SVGD kernel calls become no-ops (allocation / capability queries pass through)."""
    _keep = ("svgd_ws", "svgd_kstat", "svgd_small_supported", "svgd_fused_gram_supported")

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        attr = getattr(self._real, name)
        if name.startswith("svgd_") and name not in self._keep and callable(attr):
            return lambda *a, **k: None
        return attr


class _Done:
    def wait(self):
        return True


def multi_gpu_mode(kind, args, dist, dev, rank, world, d):
    """One exchange mode of the product's multi-GPU SVGD update (SVGDOptimizer._posterior_update, fused SGD base
    optimizer), driven exactly as SVGDOptimizer.step drives it after the backward passes: build, first contact with the
    collective, warm-up, timed blocks, then the same step re-timed with (a) the kernels and (b) the collectives
    switched off.  Every mode gets a freshly built optimizer between two barriers; an exception is NOT caught -- the
    rank exits non-zero and the launcher tears the job down (a process group is not reusable after a failed
    collective, and a rank-local fallback would leave the ranks in different collectives)."""
    import contextlib
    import beyond_deep_ensembles_amd as bde
    per = M // world
    dist.barrier()
    P0, _ = make_svgd_inputs(d, dev, 1234)                          # same seed: identical particles on every rank
    theta = torch.nn.Parameter(P0[0, :d].clone())
    rows = [P0[i, :d].clone() for i in range(M)]
    del P0
    nxt = iter(range(1, M))

    def reset():
        with torch.no_grad():
            theta.copy_(rows[next(nxt)])
    base = torch.optim.SGD([theta], lr=1e-12, momentum=0.9, nesterov=True, weight_decay=3e-4)
    kw = {"alltoall": dict(exchange="alltoall"), "pipelined": dict(exchange_chunks=args.chunks), "allgather": {}}[kind]
    opt = bde.SVGDOptimizer([theta], reset, base, particle_count=M, dataset_size=DATASET_SIZE,
                            process_group=dist.group.WORLD, fuse_base_optimizer=True, _force_exchange=(world == 1), **kw)
    del rows
    g = torch.Generator(device=dev).manual_seed(1234 + rank)          # own gradient rows differ per rank
    for i in opt._local_particles():
        opt._grad_row(i)[:d] = torch.randn(d, device=dev, generator=g) * 0.01
    loss0 = torch.zeros((), device=dev)

    def step(i=None):
        opt._posterior_update(loss0)
    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    blocks_ms = timed_blocks(step, args.steps, max(1, args.blocks), dist, dev)
    ms = median(blocks_ms)
    assert torch.isfinite(opt.particles).all()

    @contextlib.contextmanager
    def no_collectives():
        saved = (dist.all_gather_into_tensor, dist.all_to_all_single)
        fake = lambda *a, async_op=False, **k: _Done() if async_op else None
        dist.all_gather_into_tensor, dist.all_to_all_single = fake, fake
        try:
            yield
        finally:
            dist.all_gather_into_tensor, dist.all_to_all_single = saved
    real_ops = opt._ops
    opt._ops = _NoKernels(real_ops)
    exchange_only = median(timed_blocks(step, args.steps, 1, dist, dev))
    opt._ops = real_ops
    with no_collectives():
        update_only = median(timed_blocks(step, args.steps, 1, dist, dev))
    hidden = exchange_only + update_only - ms
    # bytes a rank puts on the wire per step (what the per-link xGMI bound applies to)
    ldp = pad_ld(d)
    sent = {"allgather": 4 * per * ldp * (world - 1), "pipelined": 4 * per * ldp * (world - 1),
            "alltoall": 2 * 4 * per * ldp * (world - 1) // world}[kind]
    per_rank_bytes = (12 * M + 8) * d / (world if kind == "alltoall" else 1)
    res = {"mode": kind, "step_ms": round(ms, 4), "steps_per_s": round(1e3 / ms, 2),
           "exchange_ms": round(exchange_only, 4), "update_ms": round(update_only, 4),
           "overlap": round(max(0.0, hidden) / max(1e-9, min(exchange_only, update_only)), 3),
           "ms_per_step_blocks": [round(x, 4) for x in blocks_ms],
           "bytes_sent_per_rank": int(sent), "update_bytes_per_rank": int(per_rank_bytes),
           "update_GBps_per_rank": round(per_rank_bytes / (update_only * 1e-3) / 1e9, 1),
           "_blocks": blocks_ms}
    del opt, theta, base
    torch.cuda.empty_cache()
    dist.barrier()
    return res


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps steps; the median is reported")
    ap.add_argument("--dim", type=int, default=D_RESNET50)
    ap.add_argument("--exchange", default="all", choices=["all", "allgather", "pipelined", "alltoall"],
                    help="N > 1: gradient exchange(s) of the SVGD update to time; 'all' = every mode in this one run "
                         "(headline = allgather, the exchange north_star names)")
    ap.add_argument("--chunks", type=int, default=8, help="column chunks of the pipelined all-gather")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config-extras", action="store_true",
                    help="skip extra.other_baseline_configs (smaller launches of the same kernels; PMC passes average per kernel)")
    ap.add_argument("--extras-in-process", action="store_true",
                    help="ONE process measures everything (headline, cpu baseline, `extra`) instead of the GPU-free parent with "
                         "one child per part (profiling runs: rocprofv3 then sees every kernel in THIS process's trace; a GPU "
                         "fault in an extra then costs the whole line)")
    ap.add_argument("--headline-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--extras-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from profiles/roofline_traffic.json instead of two rocprofv3 --pmc child passes in this run")
    return ap.parse_args(argv)


def traffic_child(args):
    """The child of live_traffic(): a few full-size steps of the headline's kernels (under rocprofv3 --pmc)."""
    dev_index = int(os.environ.get("BDE_BENCH_DEVICE", "0"))
    torch.cuda.set_device(dev_index)
    from beyond_deep_ensembles_amd.ops import HipOps
    dev = torch.device("cuda", dev_index)
    ops = HipOps()
    P, G = make_svgd_inputs(args.dim, dev, 1234)
    out = torch.empty_like(G)
    ws, ks = ops.svgd_ws(M, dev), ops.svgd_kstat(M, dev)
    for _ in range(4):
        ops.svgd_gram(P, args.dim, ws)
        ops.svgd_kstats(ws, M, 0.0, 1.0, DATASET_SIZE, -1.0, ks)
        ops.svgd_combine(P, G, out, args.dim, ks)
    torch.cuda.synchronize()


def extras_child(args):
    """The child of extras_in_child(): `extra` only, written to a file section by section."""
    dev_index = int(os.environ.get("BDE_BENCH_DEVICE", "0"))
    torch.cuda.set_device(dev_index)
    from beyond_deep_ensembles_amd.ops import HipOps
    dev = torch.device("cuda", dev_index)
    single_gpu_extras(HipOps(), dev, args, sink=args.extras_child)


def guarded(name, fn, *a, **k):
    """fn(*a, **k), or {"error": ...}: no optional section of the line may cost the line."""
    try:
        return fn(*a, **k)
    except (Exception, SystemExit) as e:          # noqa: BLE001 -- whatever a section raises: the line is still printed
        import traceback
        log(f"  {name} failed: {type(e).__name__}: {e}")
        return {"error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()[-1200:]}


def apply_live_traffic(res, d, dev_index) -> None:
    """roofline.traffic of ``res`` from two rocprofv3 --pmc child passes of THIS run (live_traffic); the recorded value stays
    (with the reason appended to its source) when they cannot run."""
    r = res.get("roofline")
    if not isinstance(r, dict):
        return
    live, why = live_traffic(d, dev_index)
    if live is not None:
        r["traffic"], r["traffic_source"] = live, why
    else:
        log(f"live traffic measurement skipped: {why}")
        r["traffic_source"] = (r.get("traffic_source") or "none recorded") + f" (the in-run rocprofv3 passes were skipped: {why})"


def headline(args, sink=None):
    """The process that touches the GPU for the headline: timed region, roofline, SWAG rate, the reference's op sequence on
    the same GPU, and under torch.distributed.run with one rank the forced RCCL exchange.  ``sink(res)`` (the headline child
    of the GPU-free parent, N = 1): called with everything measured so far before each optional section starts, so a fault
    there cannot cost what exists.  Without a sink (N > 1, --extras-in-process) this one process also measures the CPU
    baseline, `extra` and the PMC traffic.  Returns the result dictionary on rank 0, None on the other ranks."""
    res = None
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # BDE_BENCH_DEVICE / BDE_BENCH_BACKEND exist only to smoke-test the N > 1 code path on a 1-GPU box
    # (all ranks on one device over gloo); the driver never sets them.
    dev_index = int(os.environ.get("BDE_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # the kernel library first: HipOps() loads every code object of libbde_hip.so on this device (bde_init) BEFORE any
    # communicator thread exists (a first launch under in-flight collectives is the one ingredient of
    # the illegal-instruction queue abort that is ours to avoid); raises if the library is missing -- no fallback
    from beyond_deep_ensembles_amd.ops import HipOps
    ops = HipOps()
    dist = None
    # launched by torch.distributed.run (RANK / MASTER_ADDR set): a process group even with ONE rank, so that
    # `torchrun --nproc-per-node 1 bench.py --gpus 1` drives barriers, the MAX reduction and extra.rccl_one_rank over RCCL
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if world > 1 or launched:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("BDE_BENCH_BACKEND", "nccl")
        import datetime
        limit = datetime.timedelta(minutes=10)          # a rendezvous / collective that never completes fails, not hangs
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)
        if M % world:
            raise SystemExit(f"M={M} particles cannot be sharded over {world} ranks")

    d = args.dim
    per = M // world
    exchange_used, exchange_note = "none", None
    phases = None
    if dist is not None and os.environ.get("BDE_BENCH_BACKEND", "nccl") != "nccl":
        exchange_note = "smoke run: all ranks share one device over " + os.environ["BDE_BENCH_BACKEND"] + \
                        " (exchange times are host-staged copies, not xGMI)"

    modes_out, headline_mode, best_mode = None, None, None
    if world == 1:
        P, G = make_svgd_inputs(d, dev, 1234)
        out = torch.empty_like(G)
        ws, ks = ops.svgd_ws(M, dev), ops.svgd_kstat(M, dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

        def step(i=None):
            ops.svgd_gram(P, d, ws)
            ops.svgd_kstats(ws, M, 0.0, 1.0, DATASET_SIZE, -1.0, ks)
            if i is not None:
                ev[i][0].record()
            ops.svgd_combine(P, G, out, d, ks)
            if i is not None:
                ev[i][1].record()

        for _ in range(args.warmup):
            step()
        blocks_ms = timed_blocks(step, args.steps, max(1, args.blocks), dist, dev)
        ms_per_step = median(blocks_ms)
        combine_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps     # HIP events of the last block
        assert torch.isfinite(out[:, :d]).all()
    else:
        # every exchange mode of the product's multi-GPU update in ONE invocation; headline = north_star's all-gather
        wanted = ["allgather", "pipelined", "alltoall"] if args.exchange == "all" else [args.exchange]
        modes_out = {}
        for kind in wanted:
            modes_out[kind] = multi_gpu_mode(kind, args, dist, dev, rank, world, d)
            if rank == 0:
                log(f"[{kind}] {modes_out[kind]['step_ms']} ms/step  exchange {modes_out[kind]['exchange_ms']}  "
                    f"update {modes_out[kind]['update_ms']}")
        headline_mode = "allgather" if "allgather" in modes_out else wanted[0]
        best_mode = min(modes_out, key=lambda k: modes_out[k]["step_ms"])
        blocks_ms = modes_out[headline_mode].pop("_blocks")
        for k in modes_out:
            modes_out[k].pop("_blocks", None)
        ms_per_step = median(blocks_ms)
        exchange_used = headline_mode
        phases = modes_out[headline_mode]
        combine_ms = None

    # ---- SWAG posterior samples/s (the second half of BASELINE's metric): every rank samples independently
    # (MultiSWAG fan-out, DeepEnsemble.predict(rank=, world_size=)); aggregate = sum over ranks ("weak").
    def swag_rates():
        torch.cuda.empty_cache()
        ld = pad_ld(d)
        gsw = torch.Generator(device=dev).manual_seed(99 + rank)
        stat = torch.randn(K_SWAG + 2, ld, device=dev, generator=gsw) * 1e-3     # ring rows, mean, second moment
        stat[K_SWAG + 1] += 1e-4
        bm, bs, br = stat[K_SWAG], stat[K_SWAG + 1], stat[:K_SWAG]
        o1 = torch.empty(ld, device=dev)
        ob = torch.empty(S_SWAG, ld, device=dev)
        if dist:
            dist.barrier()
        t_single = time_loop(lambda: ops.swag_sample(bm, bs, br, 3, o1, d, seed=1, stream_id=2), 20)
        t_batch = time_loop(lambda: ops.swag_sample_batched(bm, bs, br, 3, ob, d, seed=1, stream_id0=0), 8)
        rates = torch.tensor([1.0 / t_single, S_SWAG / t_batch], device=dev, dtype=torch.float64)
        if dist:
            dist.all_reduce(rates, op=dist.ReduceOp.SUM)
        return {"samples_per_s": round(float(rates[0]), 1), "samples_per_s_batched_S30": round(float(rates[1]), 1),
                "K": K_SWAG, "D": d, "scaling": "weak (independent posterior samples on every GPU)",
                "per_sample_hbm_frac_rank0": round(4 * d * (K_SWAG + 3) / t_single / 1e9 / HBM_PEAK_GBS, 4),
                "batched_hbm_frac_rank0": round(4 * d * (K_SWAG + 2 + S_SWAG) / t_batch / 1e9 / HBM_PEAK_GBS, 4)}

    swag = None
    if d == D_RESNET50 and world > 1:
        swag = swag_rates()           # a collective inside: every rank, unguarded (an exception on one rank must end the job)

    if rank == 0:
        res = {
            "metric": "svgd_steps_per_s", "value": round(1e3 / ms_per_step, 2), "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "timing": {"blocks": len(blocks_ms), "steps_per_block": args.steps, "statistic": "median block",
                       "ms_per_step_blocks": [round(x, 4) for x in blocks_ms],
                       "ms_per_step_min": round(min(blocks_ms), 4), "ms_per_step_max": round(max(blocks_ms), 4)},
            "config": {"workload": "SVGD posterior update (svgd.py:83-89): 8 particles x ResNet-50-sized flat weights "
                                   "(iWildCam config, BASELINE configs[3] shape on N GPUs / its 1-GPU form at N=1); "
                                   + ("gram + kernel stats + combine, P and G resident in HBM" if world == 1 else
                                      "gradient exchange + kernel stats + fused update (-phi and 8 shared-state SGD "
                                      "applications) + return of the updated particles to their owners"),
                       "particles": M, "D": d, "ld": pad_ld(d), "l2_reg": 0.0, "kernel_grad_scale": 1.0,
                       "dataset_size": DATASET_SIZE, "particles_per_rank": per,
                       "exchange": exchange_used, "exchange_note": exchange_note, "best_exchange": best_mode,
                       "algorithmic_bytes_per_step": 16 * M * d if world == 1 else (12 * M + 8) * d},
            "step_hbm_frac": round((16 * M * d if world == 1 else (12 * M + 8) * d / world) / (ms_per_step * 1e-3) / 1e9
                                   / HBM_PEAK_GBS, 4),
        }
        if world == 1:
            alg_bytes = 12 * M * d                       # dominant kernel: combine reads P and G, writes out
            achieved = alg_bytes / (combine_ms * 1e-3) / 1e9
            traffic, traffic_source = None, None
            tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
            if os.path.exists(tpath) and d == D_RESNET50:
                try:
                    tj = json.load(open(tpath))
                    traffic = tj.get("svgd_combine_kernel_bytes_per_launch")
                    traffic_source = "profiles/roofline_traffic.json: " + tj.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x2 on gfx950)") \
                        + " -- a recorded PMC measurement of this kernel at this size, not re-measured in this run"
                except Exception:
                    traffic = None
            res["roofline"] = {"kernel": "svgd_combine_kernel<8,true>", "bound": "hbm", "achieved": round(achieved, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                               "traffic": traffic, "traffic_source": traffic_source,
                               "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(combine_ms, 4),
                               "avg_launch_source": "HIP events around the kernel on its stream, last timed block",
                               "served_by": "HBM + Infinity Cache: in the step the kernel runs right behind the Gram pass, "
                                            "whose last ~240 MB of particle loads are still in the 256 MB MALL "
                                            "(svgd.hip:63-68); back_to_back is the same kernel without that help"}
        else:
            # per rank the update streams (12 M + 8) D / W bytes of its slice (alltoall) or (12 M + 8) D (replicated);
            # update_ms is the step with the collectives switched off, i.e. ALL of the rank's kernels + host logic,
            # so this is a lower bound on the fused kernel's own rate
            per_rank = (12 * M + 8) * d / (world if exchange_used == "alltoall" else 1)
            ach = per_rank / (phases["update_ms"] * 1e-3) / 1e9
            res["roofline"] = {"kernel": "svgd_fused_kernel<8,0,false> (+ gram, statistics) on this rank's share", "bound": "hbm",
                               "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                               "algorithmic_bytes_per_launch": int(per_rank), "avg_launch_ms": phases["update_ms"],
                               "avg_launch_source": "the step re-timed with the collectives switched off (kernels + host logic "
                                                    "of one rank): a lower bound on the kernel's rate"}
            # every mode of this one run; the headline (value / ms_per_step) is `headline` = north_star's all-gather
            res["exchange"] = dict(modes_out, headline=headline_mode, best=best_mode,
                                   what="per mode: step_ms = median timed block; exchange_ms = the step with the SVGD kernels "
                                        "switched off (collectives + host logic); update_ms = the step with the collectives "
                                        "switched off (kernels + host logic); overlap = (exchange + update - step) / "
                                        "min(exchange, update)")
        if swag is not None:
            res["swag"] = swag
        log(f"svgd_step: {ms_per_step:.4f} ms/step (blocks {[round(x, 4) for x in blocks_ms]}) = {res['value']} steps/s; "
            f"combine {combine_ms} ms")
        if world == 1:
            # From here on nothing may cost the line: every further section is optional and guarded, and with a sink (the
            # headline child of the GPU-free parent) what exists so far is written out before each of them starts.
            one_process = sink is None
            if sink:
                sink(res)
            probe = guarded("stream_probe", stream_probe, ops, P, G, out, d, ws, ks)
            if isinstance(probe, dict) and "error" in probe:
                res["roofline"]["probe_error"] = probe["error"]
            elif probe is not None:
                r = res["roofline"]
                r["probe_GBps"] = round(probe["probe_in_step_GBps"], 1)
                r["frac_of_probe"] = round(achieved / probe["probe_in_step_GBps"], 4)
                r["probe"] = {"what": "bench_probe/probe.hip: 16 rows read + 8 rows written, the kernel's walk and "
                                      "non-temporal accesses, no arithmetic; timed by HIP events in this run, in-step = "
                                      "right behind a Gram pass over the same rows (as the kernel runs)",
                              "in_step_ms": round(probe["probe_in_step_ms"], 4),
                              "back_to_back_ms": round(probe["probe_back_to_back_ms"], 4),
                              "back_to_back_GBps": round(probe["probe_back_to_back_GBps"], 1)}
                r["back_to_back"] = {"avg_launch_ms": round(probe["combine_back_to_back_ms"], 4),
                                     "achieved": round(probe["combine_back_to_back_GBps"], 1),
                                     "frac": round(probe["combine_back_to_back_GBps"] / HBM_PEAK_GBS, 4),
                                     "frac_of_probe": round(probe["combine_back_to_back_GBps"] /
                                                            probe["probe_back_to_back_GBps"], 4)}
            del out
            if d == D_RESNET50:
                # the second half of BASELINE's metric; its kernels are device-verified, the section is guarded all the same
                res["swag"] = guarded("swag", swag_rates)
                log(f"  swag {res['swag']}")
            if sink:
                sink(res)
            if not args.no_cpu_baseline:
                if one_process:
                    log("cpu baseline ...")
                    res["cpu_baseline"] = guarded("cpu_baseline", cpu_baseline, P, G, d)
                    log(f"  {res['cpu_baseline']}")
                res["gpu_torch_baseline"] = guarded("gpu_torch_baseline", torch_gpu_baseline, P, G, d, dev)   # informational
                log(f"  gpu_torch_baseline {res['gpu_torch_baseline']}")
                if sink:
                    sink(res)
            del P, G
            torch.cuda.empty_cache()
            if not args.no_extras and d == D_RESNET50:
                if one_process:
                    log("extras ...")
                    res["extra"] = guarded("extras", single_gpu_extras, ops, dev, args)
                if dist is not None:
                    # one rank under torch.distributed.run: the product's multi-GPU update forced through the RCCL
                    # collectives (all_gather_into_tensor in place, the chunk pipeline, all_to_all_single)
                    one = {}
                    for kind in ("allgather", "pipelined", "alltoall"):
                        one[kind] = guarded(f"rccl_one_rank[{kind}]", multi_gpu_mode, kind, args, dist, dev, rank, world, d)
                        one[kind].pop("_blocks", None)
                    one["what"] = ("world size 1 over " + os.environ.get("BDE_BENCH_BACKEND", "nccl") + ": SVGDOptimizer("
                                   "process_group=WORLD, _force_exchange=True)._posterior_update -- every collective of the "
                                   "multi-GPU step executes (self-exchange, no wire); exchange_ms is launch + copy cost")
                    res["rccl_one_rank"] = one                          # (moved under `extra` when the line is assembled)
                    log(f"  rccl_one_rank {one}")
                    if sink:
                        sink(res)
            if one_process and not args.no_live_traffic:
                guarded("live_traffic", apply_live_traffic, res, d, dev_index)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    return res if rank == 0 else None


def finish_line(res):
    """`rccl_one_rank` belongs under `extra` (the headline child measured it, the extras child everything else)."""
    if isinstance(res, dict) and "rccl_one_rank" in res:
        extra = res.get("extra")
        if not isinstance(extra, dict):
            extra = res["extra"] = {} if extra is None else {"error": str(extra)}
        extra["rccl_one_rank"] = res.pop("rccl_one_rank")
    return res


def _child_env(dev_index, keep_rendezvous=False):
    drop = () if keep_rendezvous else ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                       "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env["BDE_BENCH_DEVICE"] = str(dev_index)
    return env


def headline_in_child(args, dev_index, limit_s=900):
    """The headline measured by a CHILD process (`python bench.py --headline-child FILE ...`), so that the parent never
    initialises the GPU -- on this pool a GPU-initialised process that creates further processes is what one avoids -- and so
    that a fault in one of the optional sections behind the timed region costs only that section: the child rewrites FILE
    before each of them.  Under torch.distributed.run (one rank) the child inherits the rendezvous variables and is the rank.
    -> (result dictionary or None, error string or None)."""
    import subprocess
    import tempfile
    fd, path = tempfile.mkstemp(suffix=".json", prefix="bde_bench_headline_")
    os.close(fd)
    cmd = [sys.executable, os.path.abspath(__file__), "--headline-child", path, "--gpus", "1", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--blocks", str(args.blocks), "--dim", str(args.dim), "--exchange", args.exchange,
           "--chunks", str(args.chunks)]
    for flag, on in (("--no-extras", args.no_extras), ("--no-cpu-baseline", args.no_cpu_baseline),
                     ("--no-config-extras", args.no_config_extras)):
        if on:
            cmd.append(flag)
    err = None
    try:
        proc = subprocess.Popen(cmd, env=_child_env(dev_index, keep_rendezvous=True), stdout=subprocess.DEVNULL)
        try:
            rc = proc.wait(timeout=limit_s)
            if rc != 0:
                err = f"the headline child exited with code {rc}"
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.wait()
            err = f"the headline child did not finish within {limit_s} s and was killed"
        try:
            with open(path) as f:
                res = json.load(f)
        except (OSError, ValueError):
            res = None
        return res, err
    except OSError as e:                       # the process could not be created at all
        return None, f"{type(e).__name__}: {e}"
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


def orchestrate(args):
    """N = 1, the default: a parent that NEVER touches the GPU runs the parts as sibling children, one after the other --
    the headline (timed region, roofline, SWAG rate), the `extra` sections, the two rocprofv3 --pmc passes for
    roofline.traffic -- times the CPU baseline itself (host cores only), merges what they wrote and prints the ONE line.
    Each part has its own time limit and its own failure record; the line is printed whatever happens to any of them.
    Returns the exit code (non-zero only when the headline itself is missing)."""
    dev_index = int(os.environ.get("BDE_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    d = args.dim
    res, err = headline_in_child(args, dev_index)
    if not isinstance(res, dict) or "value" not in res:
        line = {"metric": "svgd_steps_per_s", "value": None, "unit": "steps/s", "n_gpus": 1, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic", "config": {"workload": "SVGD posterior update, 8 particles x ResNet-50-sized weights"},
                "error": err or "the headline child wrote no result"}
        print(json.dumps(line), flush=True)
        return 1
    if err:
        res["headline_child_error"] = err + " after the timed region (the sections it had finished are kept)"
    if not args.no_cpu_baseline:
        log("cpu baseline ...")

        def cpu_part():
            P, G = make_svgd_inputs(d, torch.device("cpu"), 1234)       # same distribution as the device inputs, host generator
            return cpu_baseline(P, G, d)
        res["cpu_baseline"] = guarded("cpu_baseline", cpu_part)
        log(f"  {res['cpu_baseline']}")
    if not args.no_extras and d == D_RESNET50:
        log("extras ...")
        res["extra"] = guarded("extras", extras_in_child, args, dev_index)
    if not args.no_live_traffic:
        guarded("live_traffic", apply_live_traffic, res, d, dev_index)
    print(json.dumps(finish_line(res)), flush=True)
    return 0


def main():
    args = parse_args()
    if args.traffic_child:
        return traffic_child(args)
    if args.extras_child:
        return extras_child(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.headline_child:
        def sink(res):
            tmp = args.headline_child + ".tmp"
            with open(tmp, "w") as fh:
                json.dump(res, fh)
            os.replace(tmp, args.headline_child)
        headline(args, sink=sink)
        return
    if world > 1 or args.gpus > 1 or args.extras_in_process or being_profiled():
        # N > 1: every rank is already a child of torch.distributed.run and creates no process of its own.  Under rocprofv3 the
        # profiler's preloaded library has initialised the GPU in THIS process already: one process measures everything (and
        # the trace then holds every kernel of the run)
        res = headline(args)
        if res is not None:
            print(json.dumps(finish_line(res)), flush=True)
        return
    sys.exit(orchestrate(args))


if __name__ == "__main__":
    main()
