"""Kernel-level parity on a real MI355X: every C-ABI entry point (called through
ctypes, beyond_deep_ensembles_amd.ops.HipOps) against the CPU oracle and the
golden vectors captured from the imported reference.

Tolerances (fp32):
  * SWAG moment update, iVON update: separately rounded IEEE ops in the
    reference's order -> compared bit for bit (array_equal).  iVON draw:
    anchored on the fp64 evaluation of ivorn.py:108 -- ours may deviate from
    it by at most twice the reference's own fp32 deviation (floor 1 ulp).
  * SVGD: the anchor is the fp64 evaluation of the reference formula
    (tests/golden phi64).  The reference's own fp32 result deviates from it by
    err_ref; ours must stay within max(2*err_ref, 3e-6 * scale).
  * SWAG sample / Gaussian draw / KL: 2e-6 relative to the magnitude of the
    terms (different summation order / device expf, log1pf).
"""
import math

import numpy as np
import pytest
import torch

from oracle import bde_oracle as O

pytestmark = pytest.mark.gpu
# kernels that have not been green on an MI355X for the sources in this tree: their tests carry this marker and are collected
# BEHIND every test of a device-verified kernel (tests/conftest.py; the CPU-model runs of these bodies enforce the marker)
unverified = pytest.mark.device_unverified

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from beyond_deep_ensembles_amd.ops import HipOps
    return HipOps()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def flat_rows(x: torch.Tensor, ld=None):
    """[M, D] CPU tensor -> device buffer [M, ld] (ld multiple of 64), filled with NaN padding."""
    m, d = x.shape
    ld = ld or (d + 63) // 64 * 64
    buf = torch.full((m, ld), float("nan"), dtype=torch.float32, device=DEV)
    buf[:, :d] = x.to(DEV)
    return buf


def padded(x: torch.Tensor):
    d = x.numel()
    buf = torch.full(((d + 63) // 64 * 64,), float("nan"), dtype=torch.float32, device=DEV)
    buf[:d] = x.to(DEV)
    return buf


# ------------------------------------------------------------------ SVGD --
def run_svgd(ops, P, G, l2, scale, n, sign=-1.0):
    m, d = P.shape
    Pb, Gb = flat_rows(P), flat_rows(G)
    out = torch.zeros_like(Gb)
    ws, ks = ops.svgd_ws(m, DEV), ops.svgd_kstat(m, DEV)
    ops.svgd_step(Pb, Gb, out, d, l2, scale, n, sign, ws, ks)
    torch.cuda.synchronize()
    return out[:, :d].cpu(), ks.cpu()


def run_svgd_small(ops, P, G, l2, scale, n, sign=-1.0):
    """bde_svgd_step_small: the small-model kernel (two launches), an explicit choice since ABI 406."""
    m, d = P.shape
    Pb, Gb = flat_rows(P), flat_rows(G)
    out = torch.zeros_like(Gb)
    ws, ks = ops.svgd_ws(m, DEV), ops.svgd_kstat(m, DEV)
    ops.svgd_step_small(Pb, Gb, out, d, l2, scale, n, sign, ws, ks)
    torch.cuda.synchronize()
    return out[:, :d].cpu(), ks.cpu()


def run_svgd_staged(ops, P, G, l2, scale, n, sign=-1.0):
    """The three stages as separate calls (gram -> kstats -> combine): what bde_svgd_step issues in one call."""
    m, d = P.shape
    Pb, Gb = flat_rows(P), flat_rows(G)
    out = torch.zeros_like(Gb)
    ws, ks = ops.svgd_ws(m, DEV), ops.svgd_kstat(m, DEV)
    ops.svgd_gram(Pb, d, ws)
    ops.svgd_kstats(ws, m, l2, scale, n, sign, ks)
    ops.svgd_combine(Pb, Gb, out, d, ks)
    torch.cuda.synchronize()
    return out[:, :d].cpu(), ks.cpu()


def test_svgd_step_golden(ops, golden):
    g = golden("svgd_phi.npz")
    for i, (m, d, l2, scale, n, shared) in enumerate(g["cases"]):
        m, d = int(m), int(d)
        P, G = T(g[f"P_{i}"]), T(g[f"G_{i}"])
        out, ks = run_svgd(ops, P, G, float(l2), float(scale), float(n))
        phi64 = g[f"phi64_{i}"]
        ref32 = g[f"phi_{i}"].astype(np.float64)
        err_ref = np.max(np.abs(ref32 - phi64))
        scale_mag = np.max(np.abs(phi64)) + 1e-30
        err = np.max(np.abs(-out.numpy().astype(np.float64) - phi64))
        assert err <= max(2 * err_ref, 3e-6 * scale_mag), (i, m, d, err, err_ref, scale_mag)
        K = ks[:m * m].reshape(m, m).numpy()
        errK_ref = np.max(np.abs(g[f"K_{i}"].astype(np.float64) - g[f"K64_{i}"]))
        assert np.max(np.abs(K - g[f"K64_{i}"])) <= max(2 * errK_ref, 3e-6), (i, m, d)
        # bandwidth: anchored on the fp64 evaluation of svgd.py:15-18 (for shared-backbone particles the
        # reference's own fp32 value is the less accurate one, so it only sets the allowance)
        h = float(ks[2 * m * m + m])
        h64 = float(O.svgd_bandwidth(P.double()))
        assert abs(h - h64) <= max(2 * abs(float(g[f"h_{i}"]) - h64), 2e-6 * h64), (i, h, h64, float(g[f"h_{i}"]))
        # the three stages as separate calls meet the same bar (bit-identical to the one-call form)
        out3, ks3 = run_svgd_staged(ops, P, G, float(l2), float(scale), float(n))
        err3 = np.max(np.abs(-out3.numpy().astype(np.float64) - phi64))
        assert err3 <= max(2 * err_ref, 3e-6 * scale_mag), (i, m, d, err3, err_ref)
        assert abs(float(ks3[2 * m * m + m]) - h64) <= max(2 * abs(float(g[f"h_{i}"]) - h64), 2e-6 * h64)


def test_svgd_inplace_and_rbf_mode(ops, golden):
    g = golden("svgd_phi.npz")
    i = 2  # (8, 751)
    m, d, l2, scale, n, _ = g["cases"][i]
    m, d = int(m), int(d)
    P, G = T(g[f"P_{i}"]), T(g[f"G_{i}"])
    ref, _ = run_svgd(ops, P, G, float(l2), float(scale), float(n))
    Pb, Gb = flat_rows(P), flat_rows(G)
    ws, ks = ops.svgd_ws(m, DEV), ops.svgd_kstat(m, DEV)
    ops.svgd_step(Pb, Gb, Gb, d, float(l2), float(scale), float(n), -1.0, ws, ks)   # out aliases G
    assert torch.equal(Gb[:, :d].cpu(), ref)
    # rbf(): K and grad_kernel (svgd.py:14-32)
    out = torch.zeros_like(Pb)
    ops.svgd_gram(Pb, d, ws)
    ops.svgd_kstats(ws, m, 0.0, 1.0, 1.0, 1.0, ks, mode=1)
    ops.svgd_combine(Pb, None, out, d, ks)
    k64, gk64 = O.svgd_rbf(P.double())
    gk32 = g[f"gradK_{i}"].astype(np.float64)
    err_ref = np.max(np.abs(gk32 - gk64.numpy()))
    err = np.max(np.abs(out[:, :d].cpu().numpy() - gk64.numpy()))
    assert err <= max(2 * err_ref, 3e-6 * np.max(np.abs(gk64.numpy())))


def test_svgd_deterministic_and_ragged_sizes(ops):
    torch.manual_seed(0)
    for m, d in [(8, 1), (8, 3), (8, 4), (8, 127), (8, 129), (5, 4097), (16, 1000), (13, 515), (1, 77), (2, 100003)]:
        P = torch.randn(m, d) * 0.05
        G = torch.randn(m, d) * 0.01
        a, _ = run_svgd(ops, P, G, 0.01, 1.0, 5000.0)
        b, _ = run_svgd(ops, P, G, 0.01, 1.0, 5000.0)
        assert torch.equal(a, b)
        phi64 = O.svgd_phi(P.double(), G.double(), 0.01, 1.0, 5000.0).numpy()
        ref32 = O.svgd_phi(P, G, 0.01, 1.0, 5000.0).numpy().astype(np.float64)
        err_ref = np.max(np.abs(ref32 - phi64))
        err = np.max(np.abs(-a.numpy() - phi64))
        assert err <= max(2 * err_ref, 3e-6 * np.max(np.abs(phi64))), (m, d, err, err_ref)


@unverified("svgd_small")
def test_svgd_small_model_kernel(ops):
    """bde_svgd_step_small (two launches of one kernel: Gram partials; redundant statistics + combine) against the
    three-stage path and the fp64 anchor: sizes around its tile / workgroup / eligibility boundaries, the CIFAR
    ResNet-20 size of BASELINE configs 2-3, in place, and many calls on one workspace."""
    torch.manual_seed(3)
    assert ops.svgd_small_supported(8, 273_610) and ops.svgd_small_supported(8, 524_288)
    assert not ops.svgd_small_supported(8, 524_289) and not ops.svgd_small_supported(9, 1000)
    cases = [(8, 273_610), (5, 273_610), (8, 524_288), (8, 524_285), (1, 4096), (2, 127), (3, 128), (7, 129),
             (8, 32 * 128 * 3 + 5), (8, 256 * 128), (8, 256 * 128 + 1), (6, 256 * 128 * 2 - 3), (4, 70_001)]
    for m, d in cases:
        P = torch.randn(1, d) * 0.05 + torch.randn(m, d) * (0.002 if d % 2 else 0.05)     # shared-backbone-like and independent
        G = torch.randn(m, d) * 0.01
        a, ka = run_svgd_small(ops, P, G, 3e-4, 1.0, 50000.0)
        b, kb = run_svgd_staged(ops, P, G, 3e-4, 1.0, 50000.0)
        phi64 = O.svgd_phi(P.double(), G.double(), 3e-4, 1.0, 50000.0).numpy()
        ref32 = O.svgd_phi(P, G, 3e-4, 1.0, 50000.0).numpy().astype(np.float64)
        err_ref = np.max(np.abs(ref32 - phi64))
        tol = max(2 * err_ref, 3e-6 * np.max(np.abs(phi64)))
        assert np.max(np.abs(-a.numpy() - phi64)) <= tol, (m, d)
        assert np.max(np.abs(-b.numpy() - phi64)) <= tol, (m, d)
        assert torch.allclose(ka[:m * m], kb[:m * m], rtol=0, atol=2e-6), (m, d)        # K
        assert torch.allclose(a, b, rtol=0, atol=float(tol)), (m, d)


@unverified("svgd_small")
def test_svgd_small_model_kernel_repeated_calls_and_rbf(ops):
    """... in place, and repeated calls on ONE workspace beside unrelated work on another stream: same bits every time;
    the rbf mode through the same launch."""
    torch.manual_seed(4)
    m, d = 8, 273_610
    P, G = torch.randn(m, d) * 0.05, torch.randn(m, d) * 0.01
    Pb, G0 = flat_rows(P), flat_rows(G)
    ws, ks = ops.svgd_ws(m, DEV), ops.svgd_kstat(m, DEV)
    ref = torch.zeros_like(G0)
    ops.svgd_step_small(Pb, G0, ref, d, 3e-4, 1.0, 50000.0, -1.0, ws, ks)
    hog_stream, hog = torch.cuda.Stream(), torch.empty(64 << 20, device=DEV)
    for it in range(50):
        with torch.cuda.stream(hog_stream):
            hog.add_(1.0)
        Gb = G0.clone()
        ops.svgd_step_small(Pb, Gb, Gb, d, 3e-4, 1.0, 50000.0, -1.0, ws, ks)
    torch.cuda.synchronize()
    assert torch.equal(Gb[:, :d], ref[:, :d])
    # rbf mode (grad_kernel) through the same launch
    out = torch.zeros_like(Pb)
    ops.svgd_step_small(Pb, None, out, d, 0.0, 1.0, 1.0, 1.0, ws, ks, mode=1)
    k64, gk64 = O.svgd_rbf(P.double())
    k32, gk32 = O.svgd_rbf(P)
    err_ref = (gk32.double() - gk64).abs().max().item()
    assert (out[:, :d].cpu().double() - gk64).abs().max().item() <= max(2 * err_ref, 3e-6 * gk64.abs().max().item())
    # hipGraph capture of these two launches (test_svgd_step_is_graph_capturable records the streaming form)
    Gg, outg = G0.clone(), torch.zeros_like(G0)
    s = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        ops.svgd_step_small(Pb, Gg, outg, d, 3e-4, 1.0, 50000.0, -1.0, ws, ks)     # warm-up on the side stream
        s.synchronize()
        with torch.cuda.graph(graph, stream=s):
            ops.svgd_step_small(Pb, Gg, outg, d, 3e-4, 1.0, 50000.0, -1.0, ws, ks)
    outg.zero_()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(outg[:, :d], ref[:, :d])


def test_svgd_every_particle_count(ops):
    """The SVGD kernels are templates over the particle count (1 .. 16 on the single-tile path): every M once through
    -phi (against the fp64 anchor), rbf()'s grad_kernel, the fused SGD / Adam steps (== combine + apply).  The reference's
    configs use 5 particles, BASELINE 8.  (The small-model kernel's template forms: the next test.)"""
    torch.manual_seed(23)
    d = 1003
    for m in range(1, 17):
        P = torch.randn(1, d) * 0.05 + torch.randn(m, d) * 0.02
        G = torch.randn(m, d) * 0.01
        a, ks = run_svgd_staged(ops, P, G, 3e-4, 1.0, 5000.0)
        phi64 = O.svgd_phi(P.double(), G.double(), 3e-4, 1.0, 5000.0).numpy()
        ref32 = O.svgd_phi(P, G, 3e-4, 1.0, 5000.0).numpy().astype(np.float64)
        tol = max(2 * np.max(np.abs(ref32 - phi64)), 3e-6 * np.max(np.abs(phi64)))
        assert np.max(np.abs(-a.numpy() - phi64)) <= tol, m
        Pb, Gb = flat_rows(P), flat_rows(G)
        ws, kst, out = ops.svgd_ws(m, DEV), ops.svgd_kstat(m, DEV), torch.zeros_like(Pb)
        ops.svgd_gram(Pb, d, ws)
        ops.svgd_kstats(ws, m, 0.0, 1.0, 1.0, 1.0, kst, mode=1)                       # rbf(): grad_kernel, no gradients
        ops.svgd_combine(Pb, None, out, d, kst)
        _, gk64 = O.svgd_rbf(P.double())
        _, gk32 = O.svgd_rbf(P)
        err_ref = (gk32.double() - gk64).abs().max().item()
        assert (out[:, :d].cpu().double() - gk64).abs().max().item() <= max(2 * err_ref, 3e-6 * gk64.abs().max().item() + 1e-12), m
        # fused steps == statistics + combine + apply, with carried optimizer state
        for kind in ("sgd", "adam"):
            Pa, Pf = flat_rows(P), flat_rows(P)
            Pa[:, d:] = 0
            Pf[:, d:] = 0
            tmp = torch.zeros_like(Gb)
            ksa, ksf = ops.svgd_kstat(m, DEV), ops.svgd_kstat(m, DEV)
            s0a, s1a, s0f, s1f = (torch.zeros(Pa.shape[1], device=DEV) for _ in range(4))
            for it in range(2):
                ops.svgd_gram(Pa, d, ws)
                ops.svgd_kstats(ws, m, 3e-4, 1.0, 5000.0, -1.0, ksa)
                ops.svgd_combine(Pa, Gb, tmp, d, ksa)
                ops.svgd_gram(Pf, d, ws)
                ops.svgd_kstats(ws, m, 3e-4, 1.0, 5000.0, -1.0, ksf)
                if kind == "sgd":
                    ops.svgd_apply_sgd(Pa, tmp, s0a, d, 0.05, 0.9, 0.0, 3e-4, True, it == 0)
                    ops.svgd_fused_sgd(Pf, Gb, s0f, d, ksf, 0.05, 0.9, 0.0, 3e-4, True, it == 0)
                else:
                    ops.svgd_apply_adam(Pa, tmp, s0a, s1a, d, 1e-3, 0.9, 0.999, 1e-8, 1e-2, it * m)
                    ops.svgd_fused_adam(Pf, Gb, s0f, s1f, d, ksf, 1e-3, 0.9, 0.999, 1e-8, 1e-2, it * m)
                np.testing.assert_allclose(Pf[:, :d].cpu().numpy(), Pa[:, :d].cpu().numpy(), rtol=3e-6, atol=3e-7, err_msg=str((m, kind, it)))


@unverified("svgd_small")
def test_svgd_every_particle_count_small_model_kernel(ops):
    """... and for M <= 8 the small-model kernel (a template over M as well) in all its forms: -phi, rbf mode, fused SGD / Adam
    (== its -phi form + the apply kernels)."""
    torch.manual_seed(23)
    d = 1003
    for m in range(1, 9):
        P = torch.randn(1, d) * 0.05 + torch.randn(m, d) * 0.02
        G = torch.randn(m, d) * 0.01
        phi64 = O.svgd_phi(P.double(), G.double(), 3e-4, 1.0, 5000.0).numpy()
        ref32 = O.svgd_phi(P, G, 3e-4, 1.0, 5000.0).numpy().astype(np.float64)
        tol = max(2 * np.max(np.abs(ref32 - phi64)), 3e-6 * np.max(np.abs(phi64)))
        Pb, Gb = flat_rows(P), flat_rows(G)
        ws, kst = ops.svgd_ws(m, DEV), ops.svgd_kstat(m, DEV)
        _, gk64 = O.svgd_rbf(P.double())
        _, gk32 = O.svgd_rbf(P)
        err_ref = (gk32.double() - gk64).abs().max().item()
        b = torch.zeros_like(Gb)
        ops.svgd_step_small(Pb, Gb, b, d, 3e-4, 1.0, 5000.0, -1.0, ws, kst)
        assert np.max(np.abs(-b[:, :d].cpu().numpy() - phi64)) <= tol, m
        ops.svgd_step_small(Pb, None, b, d, 0.0, 1.0, 1.0, 1.0, ws, kst, mode=1)
        assert (b[:, :d].cpu().double() - gk64).abs().max().item() <= max(2 * err_ref, 3e-6 * gk64.abs().max().item() + 1e-12), m
        for kind in ("sgd", "adam"):
            Pa, Pf = flat_rows(P), flat_rows(P)
            Pa[:, d:] = 0
            Pf[:, d:] = 0
            tmp = torch.zeros_like(Gb)
            s0a, s1a, s0f, s1f = (torch.zeros(Pa.shape[1], device=DEV) for _ in range(4))
            ops.svgd_step_small(Pa, Gb, tmp, d, 3e-4, 1.0, 5000.0, -1.0, ws, kst)
            if kind == "sgd":
                ops.svgd_apply_sgd(Pa, tmp, s0a, d, 0.05, 0.9, 0.0, 3e-4, True, True)
                ops.svgd_step_small_sgd(Pf, Gb, s0f, d, 3e-4, 1.0, 5000.0, ws, kst, 0.05, 0.9, 0.0, 3e-4, True, True)
            else:
                ops.svgd_apply_adam(Pa, tmp, s0a, s1a, d, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 0)
                ops.svgd_step_small_adam(Pf, Gb, s0f, s1f, d, 3e-4, 1.0, 5000.0, ws, kst, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 0)
            np.testing.assert_allclose(Pf[:, :d].cpu().numpy(), Pa[:, :d].cpu().numpy(), rtol=3e-6, atol=3e-7, err_msg=str((m, kind)))


def test_svgd_gram_load_flavour_split_does_not_change_results(ops):
    """The Gram pass loads the head of its walk non-temporally and keeps the last `keep` bytes cacheable for the combine pass
    (svgd.hip; 240 MB by default, so the split is only ever taken at ResNet-50 size).  Forced here at small sizes through the
    tuning hook: all non-temporal, split at several points, all cacheable -- the load flavour must not change a single bit
    of the partial sums, the statistics or -phi (M = 5: the reference's particle_count, 8, 16)."""
    torch.manual_seed(17)
    try:
        # (the last size: more tiles than the capped grid has waves, so the split falls inside the waves' loops)
        for m, d in [(5, 70_001), (8, 273_610), (16, 40_000), (8, 1_200_003)]:
            P, G = torch.randn(1, d) * 0.05 + torch.randn(m, d) * 0.01, torch.randn(m, d) * 0.01
            total = 4 * m * d
            results = []
            for keep in (0, total // 7, total // 2, total - 4096, 2 * total, -1) if d < 1_000_000 else (0, total // 3, -1):
                ops.svgd_set_gram_keep_bytes(keep)
                a, ks = run_svgd_staged(ops, P, G, 3e-4, 1.0, 50000.0)
                results.append((a, ks))
            for a, ks in results[1:]:
                assert torch.equal(a, results[0][0]) and torch.equal(ks, results[0][1]), (m, d)
            phi64 = O.svgd_phi(P.double(), G.double(), 3e-4, 1.0, 50000.0).numpy()
            ref32 = O.svgd_phi(P, G, 3e-4, 1.0, 50000.0).numpy().astype(np.float64)
            tol = max(2 * np.max(np.abs(ref32 - phi64)), 3e-6 * np.max(np.abs(phi64)))
            assert np.max(np.abs(-results[0][0].numpy() - phi64)) <= tol, (m, d)
    finally:
        ops.svgd_set_gram_keep_bytes(-1)


def test_svgd_blocked_path_for_more_than_16_particles(ops):
    """17..64 particles: pair-of-groups Gram tiles -> d2 -> statistics -> chunked combine."""
    torch.manual_seed(9)
    for m, d in [(17, 515), (24, 1003), (32, 4099), (40, 130), (64, 257)]:
        P = torch.randn(m, d) * 0.05
        if m == 32:                                   # shared backbone, distinct heads
            P = (torch.randn(d) * 0.05).repeat(m, 1)
            P[:, -100:] += torch.randn(m, 100) * 0.02
        G = torch.randn(m, d) * 0.01
        a, ks = run_svgd(ops, P, G, 0.01, 1.0, 5000.0)
        b, _ = run_svgd(ops, P, G, 0.01, 1.0, 5000.0)
        assert torch.equal(a, b)
        phi64 = O.svgd_phi(P.double(), G.double(), 0.01, 1.0, 5000.0).numpy()
        ref32 = O.svgd_phi(P, G, 0.01, 1.0, 5000.0).numpy().astype(np.float64)
        err_ref = np.max(np.abs(ref32 - phi64))
        err = np.max(np.abs(-a.numpy() - phi64))
        assert err <= max(2 * err_ref, 3e-6 * np.max(np.abs(phi64))), (m, d, err, err_ref)
        k64, _ = O.svgd_rbf(P.double())
        assert np.max(np.abs(ks[:m * m].reshape(m, m).numpy() - k64.numpy())) <= 5e-6, (m, d)
        from beyond_deep_ensembles_amd.ops import BdeKernelError
        Pb, Gb = flat_rows(P), flat_rows(G)
        with pytest.raises(BdeKernelError):           # in-place is a single-tile-path feature
            ops.svgd_combine(Pb, Gb, Gb, d, ks.to(DEV))
        if m in (17, 40):                             # rbf(): K and grad_kernel (svgd.py:14-32) on the blocked path (no gradients)
            ws, kst, out = ops.svgd_ws(m, DEV), ops.svgd_kstat(m, DEV), torch.zeros_like(Pb)
            ops.svgd_gram(Pb, d, ws)
            ops.svgd_kstats(ws, m, 0.0, 1.0, 1.0, 1.0, kst, mode=1)
            ops.svgd_combine(Pb, None, out, d, kst)
            _, gk64 = O.svgd_rbf(P.double())
            _, gk32 = O.svgd_rbf(P)
            err_ref = (gk32.double() - gk64).abs().max().item()
            assert (out[:, :d].cpu().double() - gk64).abs().max().item() <= max(2 * err_ref, 3e-6 * gk64.abs().max().item()), (m, d)


def test_svgd_rejects_bad_arguments(ops):
    from beyond_deep_ensembles_amd.ops import BdeKernelError
    P = torch.zeros(65, 64, device=DEV)
    with pytest.raises(BdeKernelError):
        ops.svgd_ws(65, DEV)
    ws, ks = ops.svgd_ws(8, DEV), ops.svgd_kstat(8, DEV)
    with pytest.raises(BdeKernelError):
        ops.svgd_gram(P, 64, ws)                      # M > 64
    with pytest.raises(BdeKernelError):
        ops.svgd_gram(torch.zeros(8, 64), 64, ws)     # CPU tensor: no CPU path
    P8 = torch.zeros(8, 64, device=DEV)
    with pytest.raises(BdeKernelError):
        ops.svgd_combine(P8, None, P8, 64, ks)        # out must not alias P


def test_svgd_fused_optimizers_match_torch_shared_state(ops):
    """svgd.py:92-103 with ONE optimizer shared by all particles (Q5)."""
    torch.manual_seed(1)
    m, d = 5, 1003
    for kind in ("sgd_nesterov", "sgd_plain", "adam", "adam_wd"):
        P0 = torch.randn(m, d) * 0.1
        model_p = torch.nn.Parameter(P0[0].clone())
        if kind == "sgd_nesterov":
            opt = torch.optim.SGD([model_p], lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4)
        elif kind == "sgd_plain":
            opt = torch.optim.SGD([model_p], lr=0.05)
        elif kind == "adam":
            opt = torch.optim.Adam([model_p], lr=1e-3)
        else:
            opt = torch.optim.Adam([model_p], lr=1e-3, weight_decay=1e-2)
        rows = [P0[i].clone() for i in range(m)]
        Pb = flat_rows(P0)
        Pb[:, d:] = 0
        buf = torch.zeros(Pb.shape[1], device=DEV)
        ea, eas = torch.zeros_like(buf), torch.zeros_like(buf)
        step0 = 0
        for it in range(3):
            grads = torch.randn(m, d) * 0.01
            for i in range(m):
                model_p.grad = grads[i].clone()
                model_p.data = rows[i]
                opt.step()
            Gb = flat_rows(grads)
            if kind.startswith("sgd"):
                pg = opt.param_groups[0]
                ops.svgd_apply_sgd(Pb, Gb, buf, d, pg["lr"], pg["momentum"], pg["dampening"], pg["weight_decay"],
                                   pg["nesterov"], first=(it == 0))
            else:
                pg = opt.param_groups[0]
                ops.svgd_apply_adam(Pb, Gb, ea, eas, d, pg["lr"], pg["betas"][0], pg["betas"][1], pg["eps"],
                                    pg["weight_decay"], step0)
                step0 += m
            want = torch.stack(rows)
            got = Pb[:, :d].cpu()
            assert torch.allclose(got, want, rtol=5e-6, atol=5e-7), (kind, it, (got - want).abs().max())


def test_svgd_fused_equals_combine_plus_apply(ops):
    """bde_svgd_fused_* == bde_svgd_combine followed by bde_svgd_apply_* (svgd.py:86-103), and the Gram
    partials it leaves for the next step give the same kernel statistics as a fresh bde_svgd_gram."""
    torch.manual_seed(5)
    for m, d in [(8, 4099), (5, 1003), (3, 17), (12, 515)]:
        P0 = torch.randn(m, d) * 0.05
        G0 = torch.randn(m, d) * 0.01
        for kind in ("sgd", "adam"):
            Pa, Pb, Gb = flat_rows(P0), flat_rows(P0), flat_rows(G0)
            tmp = torch.zeros_like(Gb)
            ws, ks = ops.svgd_ws(m, DEV), ops.svgd_kstat(m, DEV)
            wsn = ops.svgd_ws(m, DEV) if m <= 8 else None
            s0a, s1a, s0b, s1b = (torch.zeros(Pa.shape[1], device=DEV) for _ in range(4))
            for it in range(3):
                Gb[:, :d] = (G0 * (1 + it)).to(DEV)
                ops.svgd_gram(Pa, d, ws)
                ops.svgd_kstats(ws, m, 0.01, 1.0, 500.0, -1.0, ks)
                ksa = ks.clone()
                ops.svgd_combine(Pa, Gb, tmp, d, ks)
                if kind == "sgd":
                    ops.svgd_apply_sgd(Pa, tmp, s0a, d, 0.05, 0.9, 0.0, 3e-4, True, it == 0)
                else:
                    ops.svgd_apply_adam(Pa, tmp, s0a, s1a, d, 1e-3, 0.9, 0.999, 1e-8, 1e-2, it * m)
                if it == 0 or wsn is None:
                    ops.svgd_gram(Pb, d, ws)
                    ops.svgd_kstats(ws, m, 0.01, 1.0, 500.0, -1.0, ks)
                else:                       # statistics from the Gram partials the previous fused call left
                    ops.svgd_kstats(wsn, m, 0.01, 1.0, 500.0, -1.0, ks)
                    np.testing.assert_allclose(ks[:m * m].cpu().numpy(), ksa[:m * m].cpu().numpy(), rtol=2e-5, atol=1e-7)
                if kind == "sgd":
                    ops.svgd_fused_sgd(Pb, Gb, s0b, d, ks, 0.05, 0.9, 0.0, 3e-4, True, it == 0, ws_next=wsn)
                else:
                    ops.svgd_fused_adam(Pb, Gb, s0b, s1b, d, ks, 1e-3, 0.9, 0.999, 1e-8, 1e-2, it * m, ws_next=wsn)
                np.testing.assert_allclose(Pb[:, :d].cpu().numpy(), Pa[:, :d].cpu().numpy(), rtol=3e-6, atol=3e-7)
                np.testing.assert_allclose(s0b[:d].cpu().numpy(), s0a[:d].cpu().numpy(), rtol=1e-5, atol=1e-7)


@unverified("svgd_small")
def test_svgd_small_model_fused_step(ops):
    """bde_svgd_step_small_sgd / _adam (the whole SVGDOptimizer.step minus forward/backward by the small-model kernel) ==
    bde_svgd_step_small followed by bde_svgd_apply_* (svgd.py:86-103), over several steps with carried state, at
    ragged sizes and the CIFAR ResNet-20 size."""
    torch.manual_seed(15)
    for m, d in [(8, 273_610), (5, 4099), (3, 17), (8, 256 * 128 + 3), (1, 1000)]:
        P0 = torch.randn(1, d) * 0.05 + torch.randn(m, d) * 0.01
        G0 = torch.randn(m, d) * 0.01
        for kind in ("sgd", "sgd_plain", "adam"):
            Pa, Pb, Gb = flat_rows(P0), flat_rows(P0), flat_rows(G0)
            Pa[:, d:] = 0
            Pb[:, d:] = 0
            tmp = torch.zeros_like(Gb)
            ws, ks, ksb = ops.svgd_ws(m, DEV), ops.svgd_kstat(m, DEV), ops.svgd_kstat(m, DEV)
            s0a, s1a, s0b, s1b = (torch.zeros(Pa.shape[1], device=DEV) for _ in range(4))
            for it in range(3):
                Gb[:, :d] = (G0 * (1 + it)).to(DEV)
                ops.svgd_step_small(Pa, Gb, tmp, d, 0.01, 1.0, 500.0, -1.0, ws, ks)
                if kind == "sgd":
                    ops.svgd_apply_sgd(Pa, tmp, s0a, d, 0.05, 0.9, 0.0, 3e-4, True, it == 0)
                    ops.svgd_step_small_sgd(Pb, Gb, s0b, d, 0.01, 1.0, 500.0, ws, ksb, 0.05, 0.9, 0.0, 3e-4, True, it == 0)
                elif kind == "sgd_plain":              # no momentum: the state buffer is never touched
                    ops.svgd_apply_sgd(Pa, tmp, s0a, d, 0.05, 0.0, 0.0, 0.0, False, it == 0)
                    ops.svgd_step_small_sgd(Pb, Gb, s0b, d, 0.01, 1.0, 500.0, ws, ksb, 0.05, 0.0, 0.0, 0.0, False, it == 0)
                else:
                    ops.svgd_apply_adam(Pa, tmp, s0a, s1a, d, 1e-3, 0.9, 0.999, 1e-8, 1e-2, it * m)
                    ops.svgd_step_small_adam(Pb, Gb, s0b, s1b, d, 0.01, 1.0, 500.0, ws, ksb, 1e-3, 0.9, 0.999, 1e-8, 1e-2, it * m)
                assert torch.equal(ks[:m * m], ksb[:m * m]), (m, d, kind, it)          # same statistics, same bits
                np.testing.assert_allclose(Pb[:, :d].cpu().numpy(), Pa[:, :d].cpu().numpy(), rtol=3e-6, atol=3e-7)
                np.testing.assert_allclose(s0b[:d].cpu().numpy(), s0a[:d].cpu().numpy(), rtol=1e-5, atol=1e-7)
                np.testing.assert_allclose(s1b[:d].cpu().numpy(), s1a[:d].cpu().numpy(), rtol=1e-5, atol=1e-9)
            assert torch.equal(Pb[:, d:], torch.zeros_like(Pb[:, d:]))                  # padding untouched


def test_svgd_segmented_gradients_equal_flat_rows(ops):
    """The *_seg entry points read every particle's gradients from the tensors autograd produced (per-tensor
    allocations; svgd.py:129-133's clones removed) and must give the BITS of the flat-row entry points fed with the
    same gradients packed into G [M, ld]: combine, fused SGD / Adam (incl. the next step's Gram partials), and the
    one-launch packer.  Tensor sizes straddle every boundary: 1 element, numel % 4 != 0, exactly one chunk (1024),
    several chunks, and gradients the kernels cannot read in place (missing, strided, unaligned) that the collector
    routes through the flat row."""
    from beyond_deep_ensembles_amd.algo import FlatLayout, collect_grads
    torch.manual_seed(33)
    sizes = [(7,), (1,), (1024,), (33, 31), (4100,), (2, 3, 5), (1027,), (64,), (3000,)]
    for m in range(1, 17):                                     # every instantiation of the per-particle-count templates
        params = [torch.nn.Parameter(torch.randn(s, device=DEV) * 0.05) for s in sizes]
        lay = FlatLayout(params, align=4)
        d, ld = lay.d, lay.ld
        assert lay.padded and d % 4 == 0
        P = torch.zeros(m, ld, device=DEV)
        for i in range(m):
            for v, p in zip(lay.views(P[i]), params):
                v.copy_(p.detach() + 0.01 * torch.randn_like(p))
        Gflat = torch.zeros(m, ld, device=DEV)
        Gfall = torch.zeros(m, ld, device=DEV)                 # the rows the collector falls back to
        seg = ops.seg_table(lay.offsets, lay.numels, m, DEV)
        host = seg.staging()
        retained = []
        big = torch.randn(5000, device=DEV)
        for j in range(m):
            for k, p in enumerate(params):
                kind = (j + k) % 7
                if kind == 0:
                    p.grad = None                                           # missing -> zeros
                elif kind == 1:
                    p.grad = big[1:1 + p.numel()].view(p.shape) * 1.0       # fresh, aligned
                elif kind == 2:
                    p.grad = big[3:3 + p.numel()].view(p.shape)             # a view at an unaligned address
                elif kind == 3 and p.dim() == 2:
                    p.grad = (torch.randn(p.shape[1], p.shape[0], device=DEV) * 0.01).t()   # strided
                else:
                    p.grad = torch.randn_like(p) * 0.01
            for v, p in zip(lay.views(Gflat[j]), params):
                if p.grad is not None:
                    v.copy_(p.grad)
            retained.append(collect_grads(params, lay.views(Gfall[j]), host, j, m))
        assert any(len(r) < len(params) for r in retained) and all(len(r) > 0 for r in retained)
        seg.upload()
        # the packer
        ops.svgd_gather_seg(Gfall, seg, 0, m)
        torch.cuda.synchronize()
        assert torch.equal(Gfall[:, :d], Gflat[:, :d]), m
        # statistics once
        ws, ks = ops.svgd_ws(m, DEV), ops.svgd_kstat(m, DEV)
        ops.svgd_gram(P, d, ws)
        ops.svgd_kstats(ws, m, 0.01, 1.0, 500.0, -1.0, ks)
        # re-collect: the packer made every row complete, so point the table at a FRESH fallback buffer again
        Gfall2 = torch.zeros(m, ld, device=DEV)
        host = seg.staging()
        retained = []
        for j in range(m):
            for k, p in enumerate(params):
                v = lay.views(Gflat[j])[k]
                kind = (j + k) % 7
                p.grad = None if kind == 0 else (v.clone() if kind != 2 else torch.cat([v.new_zeros(1), v.reshape(-1)])[1:].view(p.shape))
            retained.append(collect_grads(params, lay.views(Gfall2[j]), host, j, m))
        seg.upload()
        out_flat, out_seg = torch.zeros(m, ld, device=DEV), torch.zeros(m, ld, device=DEV)
        ops.svgd_combine(P, Gflat, out_flat, d, ks)
        ops.svgd_combine_seg(P, seg, out_seg, d, ks)
        assert torch.equal(out_seg, out_flat), m
        # in place on the fallback rows (what the unfused shell does: out = the flat gradient rows)
        ops.svgd_combine_seg(P, seg, Gfall2, d, ks)
        assert torch.equal(Gfall2[:, :d], out_flat[:, :d]), m
        next_gram = ops.svgd_fused_gram_supported(m)           # the next step's Gram partials ride along for M <= 8
        # fused SGD / Adam: the table must be rebuilt because the in-place combine consumed the fallback rows
        for kind in ("sgd", "adam"):
            Gfall3 = torch.zeros(m, ld, device=DEV)
            host = seg.staging()
            retained = []
            for j in range(m):
                for k, p in enumerate(params):
                    v = lay.views(Gflat[j])[k]
                    p.grad = None if (j + k) % 7 == 0 else v.clone()
                retained.append(collect_grads(params, lay.views(Gfall3[j]), host, j, m))
            seg.upload()
            Pa, Pb = P.clone(), P.clone()
            s0a, s0b, s1a, s1b = (torch.zeros(ld, device=DEV) for _ in range(4))
            wa, wb = (ops.svgd_ws(m, DEV), ops.svgd_ws(m, DEV)) if next_gram else (None, None)
            for it in range(2):
                if kind == "sgd":
                    ops.svgd_fused_sgd(Pa, Gflat, s0a, d, ks, 0.05, 0.9, 0.0, 3e-4, True, it == 0, ws_next=wa)
                    ops.svgd_fused_sgd_seg(Pb, seg, s0b, d, ks, 0.05, 0.9, 0.0, 3e-4, True, it == 0, ws_next=wb)
                else:
                    ops.svgd_fused_adam(Pa, Gflat, s0a, s1a, d, ks, 1e-3, 0.9, 0.999, 1e-8, 1e-2, it * m, ws_next=wa)
                    ops.svgd_fused_adam_seg(Pb, seg, s0b, s1b, d, ks, 1e-3, 0.9, 0.999, 1e-8, 1e-2, it * m, ws_next=wb)
            if next_gram:                                       # ... and a step without the riding Gram partials
                if kind == "sgd":
                    ops.svgd_fused_sgd(Pa, Gflat, s0a, d, ks, 0.05, 0.9, 0.0, 3e-4, True, False)
                    ops.svgd_fused_sgd_seg(Pb, seg, s0b, d, ks, 0.05, 0.9, 0.0, 3e-4, True, False)
                else:
                    ops.svgd_fused_adam(Pa, Gflat, s0a, s1a, d, ks, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 2 * m)
                    ops.svgd_fused_adam_seg(Pb, seg, s0b, s1b, d, ks, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 2 * m)
            torch.cuda.synchronize()
            assert torch.equal(Pa, Pb) and torch.equal(s0a, s0b) and torch.equal(s1a, s1b), (m, kind)
            # the padding columns of the particles and of the state stay zero
            pad = torch.ones(ld, dtype=torch.bool, device=DEV)
            pad[lay.valid_index(DEV)] = False
            assert float(Pb[:, pad].abs().max()) == 0.0 and float(s0b[pad].abs().max()) == 0.0
            if not next_gram:
                continue
            ka, kb = ops.svgd_kstat(m, DEV), ops.svgd_kstat(m, DEV)
            ops.svgd_kstats(wa, m, 0.01, 1.0, 500.0, -1.0, ka)
            ops.svgd_kstats(wb, m, 0.01, 1.0, 500.0, -1.0, kb)
            # Gram partials of the updated particles: the two kernels partition the columns differently (float4 columns
            # vs chunks), so the partial sums differ in the last bits, the statistics agree to rounding
            assert torch.allclose(ka[:m * m], kb[:m * m], rtol=0, atol=2e-6), (m, kind)


# ------------------------------------------------------------------ SWAG --
def test_swag_batched_sampler_both_kernels_equal_single_samples(ops):
    """bde_swag_sample_batched runs the LDS-DMA pipelined kernel for K <= 20 ring rows and the register kernel above;
    either must give the BITS of S calls of bde_swag_sample (same FMA order per element, same Philox streams), for
    sizes below one 128-parameter tile, at tile boundaries, with a D % 4 tail and a partial last tile, odd and even K,
    K = 1, and S from 1 to the maximum of 32; with in-kernel and with supplied noise; nothing is written past D."""
    torch.manual_seed(41)
    # (the last size: more 128-parameter tiles than the grid has waves -- 3072 workgroups x 4 -- so that waves walk SEVERAL
    # tiles: the software pipeline of the LDS-DMA kernel and the grid-stride loop of the register kernel)
    for d in (3, 4, 127, 128, 129, 4099, 70_001, 3_300_003):
        ld = (d + 63) // 64 * 64
        for k, s_n in ((1, 2), (5, 7), (19, 30), (20, 32), (21, 3), (30, 30)) if d < 1_000_000 else ((20, 3), (21, 2)):
            stat = torch.zeros(k + 2, ld, device=DEV)
            stat[:k, :d] = torch.randn(k, d, device=DEV) * 1e-2
            stat[k, :d] = torch.randn(d, device=DEV) * 0.05
            stat[k + 1, :d] = stat[k, :d] ** 2 + torch.rand(d, device=DEV) * 1e-3
            ring, mean, sq = stat[:k], stat[k], stat[k + 1]
            head = k // 2
            out = torch.full((s_n, ld), 7.0, device=DEV)
            ops.swag_sample_batched(mean, sq, ring, head, out, d, seed=3, stream_id0=11)
            one = torch.zeros(ld, device=DEV)
            for s in range(s_n):
                ops.swag_sample(mean, sq, ring, head, one, d, seed=3, stream_id=11 + s)
                assert torch.equal(out[s, :d], one[:d]), (d, k, s)
            assert ld == d or float((out[:, d:] - 7.0).abs().max()) == 0.0, (d, k)
            ewb, edb = torch.randn(s_n, k, device=DEV), torch.zeros(s_n, ld, device=DEV)
            edb[:, :d] = torch.randn(s_n, d, device=DEV)
            ops.swag_sample_batched(mean, sq, ring, head, out, d, eps_w=ewb, eps_d=edb)
            for s in (0, s_n - 1):
                ops.swag_sample(mean, sq, ring, head, one, d, eps_w=ewb[s], eps_d=edb[s])
                assert torch.equal(out[s, :d], one[:d]), (d, k, s)


def test_swag_update_bit_exact(ops):
    torch.manual_seed(2)
    for d in (1, 5, 64, 1027, 100003):
        theta0 = torch.randn(d) * 0.05
        st = O.swag_init(theta0, 4)
        mean, sq = padded(st.mean), padded(st.sq_weights)
        dev = torch.zeros(4, mean.numel(), device=DEV)
        head = 0
        for n in range(1, 8):
            theta = theta0 + torch.randn(d) * 1e-3 * n
            st.updates = n
            O.swag_moment_update(st, theta)
            ops.swag_update(padded(theta), mean, sq, dev[head], n, d)
            head = (head + 1) % 4
            assert torch.equal(mean[:d].cpu(), st.mean)
            assert torch.equal(sq[:d].cpu(), st.sq_weights)
            # ring -> reference layout: logical column c = physical row (head + c) % K
            logical = torch.stack([dev[(head + c) % 4, :d].cpu() for c in range(4)], dim=1)
            assert torch.equal(logical, st.deviations)


def test_streaming_kernels_walk_several_grid_passes(ops):
    """Sizes beyond one pass of the capped grids (2048 workgroups x 256 threads x float4 = 2,097,152 elements): every
    thread takes several trips of its grid-stride loop.  The other tests stay below that; bench.py runs 23.9 M elements
    but only times.  SWAG moments bit-exact against the oracle, the samplers / draws against their formulas."""
    torch.manual_seed(6)
    d, k = 2_300_003, 3
    theta0 = torch.randn(d) * 0.05
    st = O.swag_init(theta0, k)
    mean, sq = padded(st.mean), padded(st.sq_weights)
    ring = torch.zeros(k, mean.numel(), device=DEV)
    head = 0
    for n in range(1, 5):
        theta = theta0 + torch.randn(d) * 1e-3 * n
        st.updates = n
        O.swag_moment_update(st, theta)
        ops.swag_update(padded(theta), mean, sq, ring[head], n, d)
        head = (head + 1) % k
    assert torch.equal(mean[:d].cpu(), st.mean) and torch.equal(sq[:d].cpu(), st.sq_weights)
    logical = torch.stack([ring[(head + c) % k, :d].cpu() for c in range(k)], dim=1)
    assert torch.equal(logical, st.deviations)
    eps_w, eps_d = O.swag_draw_noise(k, d)
    want64 = (st.mean.double() + (st.deviations.double() / math.sqrt(2 * (k - 1))) @ eps_w.double()
              + (0.5 * (torch.relu(st.sq_weights.double() - st.mean.double() ** 2) + 1e-6)).sqrt() * eps_d.double())
    ref32 = O.swag_sample(st.mean, st.sq_weights, st.deviations, eps_w, eps_d)
    out = torch.zeros_like(mean)
    ops.swag_sample(mean, sq, ring, head, out, d, eps_w=eps_w.to(DEV), eps_d=padded(eps_d))
    err, err_ref = (out[:d].cpu().double() - want64).abs().max().item(), (ref32.double() - want64).abs().max().item()
    assert err <= max(2 * err_ref, 1e-7), (err, err_ref)
    # in-kernel noise: the sample is the supplied-noise sample of the Philox stream's normals, element for element
    ew, ed = torch.zeros(k, device=DEV), torch.zeros_like(mean)
    ops.philox_normal(9, 4, eps_w=ew, eps_d=ed, d=d, rounds=ops.swag_philox_rounds)
    a, b = torch.zeros_like(mean), torch.zeros_like(mean)
    ops.swag_sample(mean, sq, ring, head, a, d, seed=9, stream_id=4)
    ops.swag_sample(mean, sq, ring, head, b, d, eps_w=ew, eps_d=ed)
    assert torch.allclose(a[:d], b[:d], rtol=1e-6, atol=1e-7)
    # Gaussian draw and the local-reparameterisation epilogue
    rho, eps = torch.randn(mean.numel(), device=DEV) - 2.0, torch.randn(mean.numel(), device=DEV)
    w = torch.zeros_like(mean)
    ops.gauss_draw_fwd(mean, rho, w, d, eps=eps)
    assert torch.allclose(w[:d], (mean + torch.nn.functional.softplus(rho) * eps)[:d], rtol=2e-6, atol=1e-6)
    var = torch.rand(mean.numel(), device=DEV) + 1e-3
    ops.local_reparam_fwd(mean, var, w, d, eps=eps)
    assert torch.allclose(w[:d], (mean + var.sqrt() * eps)[:d], rtol=2e-6, atol=1e-6)
    assert float(w[d:].abs().max()) == 0.0 if w.numel() > d else True          # nothing written past d


def test_swag_sample_golden_and_oracle(ops, golden):
    g = golden("swag_stats.npz")
    for ci in range(len(g["cases"])):
        mean, sq, devDK = T(g[f"mean_{ci}"]), T(g[f"sq_{ci}"]), T(g[f"dev_{ci}"])
        d, k = devDK.shape
        for head in (0, 2):
            ring = torch.zeros(k, (d + 63) // 64 * 64, device=DEV)
            for c in range(k):
                ring[(head + c) % k, :d] = devDK[:, c].to(DEV)
            out = torch.zeros((d + 63) // 64 * 64, device=DEV)
            for s in range(3):
                ops.swag_sample(padded(mean), padded(sq), ring, head, out, d, eps_w=T(g[f"eps_w_{ci}"][s]).to(DEV),
                                eps_d=padded(T(g[f"eps_d_{ci}"][s])))
                want = g[f"samples_{ci}"][s]
                mag = np.abs(g[f"mean_{ci}"]) + np.abs(want) + 1e-3
                assert np.max(np.abs(out[:d].cpu().numpy() - want) / mag) < 2e-6


def test_swag_sample_large_vs_fp64(ops):
    torch.manual_seed(3)
    d, k = 50021, 20
    theta = torch.randn(d) * 0.05
    st = O.swag_init(theta, k)
    for n in range(1, 26):
        theta = theta + torch.randn(d) * 1e-3
        st.updates = n
        O.swag_moment_update(st, theta)
    eps_w, eps_d = O.swag_draw_noise(k, d)
    want64 = (st.mean.double() + (st.deviations.double() / math.sqrt(2 * (k - 1))) @ eps_w.double()
              + (0.5 * (torch.relu(st.sq_weights.double() - st.mean.double() ** 2) + 1e-6)).sqrt() * eps_d.double())
    ref32 = O.swag_sample(st.mean, st.sq_weights, st.deviations, eps_w, eps_d)
    ld = (d + 63) // 64 * 64
    ring = torch.zeros(k, ld, device=DEV)
    head = 7
    for c in range(k):
        ring[(head + c) % k, :d] = st.deviations[:, c].to(DEV)
    out = torch.zeros(ld, device=DEV)
    ops.swag_sample(padded(st.mean), padded(st.sq_weights), ring, head, out, d, eps_w=eps_w.to(DEV), eps_d=padded(eps_d))
    err = (out[:d].cpu().double() - want64).abs().max().item()
    err_ref = (ref32.double() - want64).abs().max().item()
    assert err <= max(2 * err_ref, 1e-7), (err, err_ref)
    # batched == unbatched on the same noise
    S = 5
    ew = torch.randn(S, k)
    ed = torch.randn(S, d)
    edb = torch.zeros(S, ld, device=DEV)
    edb[:, :d] = ed.to(DEV)
    outb = torch.zeros(S, ld, device=DEV)
    ops.swag_sample_batched(padded(st.mean), padded(st.sq_weights), ring, head, outb, d, eps_w=ew.to(DEV), eps_d=edb)
    for s in range(S):
        ops.swag_sample(padded(st.mean), padded(st.sq_weights), ring, head, out, d, eps_w=ew[s].to(DEV), eps_d=edb[s])
        assert torch.allclose(outb[s, :d], out[:d], rtol=1e-6, atol=1e-7), s


def test_philox_streams(ops):
    d, k = 1 << 20, 20
    ld = d
    a = torch.zeros(d, device=DEV)
    b = torch.zeros(d, device=DEV)
    w = torch.zeros(k, device=DEV)
    ops.philox_normal(1234, 0, eps_w=w, eps_d=a)
    ops.philox_normal(1234, 1, eps_d=b)
    assert torch.isfinite(a).all() and torch.isfinite(w).all()
    assert abs(a.mean().item()) < 5e-3 and abs(a.var().item() - 1.0) < 5e-3
    assert abs((a * a * a).mean().item()) < 2e-2                       # skewness ~ 0
    assert abs((a ** 4).mean().item() - 3.0) < 5e-2                    # kurtosis ~ 3
    assert abs((a * b).mean().item()) < 5e-3                           # streams uncorrelated
    assert abs((a[1:] * a[:-1]).mean().item()) < 5e-3                  # neighbours uncorrelated
    c = torch.zeros(d, device=DEV)
    ops.philox_normal(1234, 0, eps_d=c)
    assert torch.equal(a, c)                                           # pure function of (seed, stream, index)
    # the sampling kernel's in-kernel noise IS this stream: RNG mode == supplied-noise mode, bit for bit
    dd, kk = 10007, 6
    ldd = (dd + 63) // 64 * 64
    mean, sq = torch.randn(ldd, device=DEV) * 0.05, torch.rand(ldd, device=DEV)
    ring = torch.randn(kk, ldd, device=DEV) * 1e-3
    o1, o2 = torch.zeros(ldd, device=DEV), torch.zeros(ldd, device=DEV)
    ew, ed = torch.zeros(kk, device=DEV), torch.zeros(ldd, device=DEV)
    assert ops.swag_philox_rounds == 7
    ops.philox_normal(99, 5, eps_w=ew, eps_d=ed, d=dd, rounds=ops.swag_philox_rounds)    # the samplers' 7-round streams
    ops.swag_sample(mean, sq, ring, 3, o1, dd, seed=99, stream_id=5)
    ops.swag_sample(mean, sq, ring, 3, o2, dd, eps_w=ew, eps_d=ed)
    assert torch.equal(o1[:dd], o2[:dd])
    # batched RNG mode == S unbatched calls with stream ids stream0 + s
    S = 4
    ob = torch.zeros(S, ldd, device=DEV)
    ops.swag_sample_batched(mean, sq, ring, 3, ob, dd, seed=99, stream_id0=5)
    for s in range(S):
        ops.swag_sample(mean, sq, ring, 3, o1, dd, seed=99, stream_id=5 + s)
        assert torch.allclose(ob[s, :dd], o1[:dd], rtol=1e-6, atol=1e-7)


# ----------------------------------------------------------------- Gauss --
def test_gauss_draw_kl_golden(ops, golden):
    g = golden("bbb.npz")
    mean, rho, eps = T(g["a_mean"]), T(g["a_rho"]), T(g["a_eps"])
    n = mean.numel()
    mb, rb, eb = padded(mean), padded(rho), padded(eps)
    w = torch.zeros_like(mb)
    ops.gauss_draw_fwd(mb, rb, w, n, eps=eb)
    assert np.max(np.abs(w[:n].cpu().numpy() - g["a_sample"])) < 2e-6 * (np.abs(g["a_sample"]).max() + 1)
    gm, gr = torch.zeros_like(mb), torch.zeros_like(mb)
    ops.gauss_draw_bwd(padded(T(g["a_gout"])), rb, gm, gr, n, eps=eb)
    assert np.array_equal(gm[:n].cpu().numpy(), g["a_gmean"])
    np.testing.assert_allclose(gr[:n].cpu().numpy(), g["a_grho"], rtol=3e-6, atol=1e-9)
    # accumulate mode adds on top
    ops.gauss_draw_bwd(padded(T(g["a_gout"])), rb, gm, gr, n, eps=eb, accumulate=True)
    np.testing.assert_allclose(gr[:n].cpu().numpy(), 2 * g["a_grho"], rtol=3e-6, atol=1e-9)
    ws = ops.reduce_ws(DEV)
    for pi, (mu, sigma) in enumerate(g["a_priors"]):
        si, mi = pi // 2, pi % 2
        kl = torch.zeros(1, device=DEV)
        gm.zero_(); gr.zero_()
        ops.gauss_kl(mb, rb, float(mu), float(sigma), n, ws, kl_out=kl, gmean=gm, grho=gr, grad_scale=0.25)
        want = float(g[f"a_kl_{si}_{mi}"])
        assert abs(kl.item() - want) <= 3e-6 * abs(want), (kl.item(), want)
        np.testing.assert_allclose(gm[:n].cpu().numpy(), 0.25 * g[f"a_kl_gmean_{si}_{mi}"], rtol=3e-6, atol=1e-9)
        s = O.gauss_std(rho).numpy()
        scale = 0.25 * (1.0 / s + s / float(sigma) ** 2)
        assert np.max(np.abs(gr[:n].cpu().numpy() - 0.25 * g[f"a_kl_grho_{si}_{mi}"]) / scale) < 3e-6
        # value-only call and device-side scale factor
        kl2 = torch.zeros(1, device=DEV)
        ops.gauss_kl(mb, rb, float(mu), float(sigma), n, ws, kl_out=kl2)
        assert kl2.item() == kl.item()
        gm2, gr2 = torch.zeros_like(mb), torch.zeros_like(mb)
        ops.gauss_kl(mb, rb, float(mu), float(sigma), n, ws, gmean=gm2, grho=gr2, grad_scale=0.125,
                     grad_scale_dev=torch.full((1,), 2.0, device=DEV))
        assert torch.equal(gm2[:n], gm[:n]) and torch.equal(gr2[:n], gr[:n])


def test_gauss_kl_large_and_l2(ops):
    torch.manual_seed(4)
    n = 300007
    mean, rho = torch.randn(n) * 0.1, torch.randn(n) - 3
    ws = ops.reduce_ws(DEV)
    kl = torch.zeros(1, device=DEV)
    ops.gauss_kl(padded(mean), padded(rho), 0.0, 1.0, n, ws, kl_out=kl)
    want = O.gauss_kl(mean.double(), rho.double(), 0.0, 1.0).item()
    assert abs(kl.item() - want) <= 2e-6 * abs(want)
    val = torch.zeros(1, device=DEV)
    g = torch.zeros((n + 63) // 64 * 64, device=DEV)
    ops.l2(padded(mean), 0.3, n, ws, val_out=val, g=g, grad_scale=0.5)
    assert abs(val.item() - O.l2_term(mean.double(), 0.3).item()) <= 2e-6 * val.item()
    np.testing.assert_allclose(g[:n].cpu().numpy(), (0.5 * 0.3 * mean).numpy(), rtol=1e-6, atol=1e-12)


def test_mixture_prior_kernel(ops):
    """bde_mixture_nll against autograd of the reference expression (bbb.py:31-37) evaluated by the oracle: value
    and gradient, both clamp regimes (|x| small: density clamped at 0 from above; |x| large: clamped at -23)."""
    torch.manual_seed(12)
    ws = ops.reduce_ws(DEV)
    for (pi, s1, s2), n in [((0.5, 1.0, 0.05), 100003), ((0.25, 2.0, 0.002), 4099), ((0.9, 0.5, 0.5), 17)]:
        x = torch.cat([torch.randn(n // 2) * 0.1, torch.randn(n - n // 2) * 3.0])
        x[:5] = torch.tensor([0.0, 1e-4, -7.0, 30.0, -0.3])
        want = O.mixture_nll(x.double(), pi, s1, s2).item()
        want32 = O.mixture_nll(x, pi, s1, s2).item()
        g32 = O.mixture_nll_grad(x, pi, s1, s2)
        xb = padded(x)
        val, gm = torch.zeros(1, device=DEV), torch.zeros_like(xb)
        ops.mixture_nll(xb, pi, s1, s2, n, ws, val_out=val, gmean=gm, grad_scale=0.5)
        assert abs(val.item() - want) <= max(2 * abs(want32 - want), 3e-6 * abs(want)), (pi, s1, s2)
        scale = g32.abs().max().item()
        assert (gm[:n].cpu() - 0.5 * g32).abs().max().item() <= 3e-6 * scale, (pi, s1, s2)
        gm2 = gm.clone()
        ops.mixture_nll(xb, pi, s1, s2, n, ws, gmean=gm2, grad_scale=0.5, accumulate=True)
        assert torch.allclose(gm2[:n], 2 * gm[:n], rtol=1e-6, atol=1e-7 * scale)


def test_gauss_draw_philox(ops):
    n = 100003
    mean, rho = torch.randn(n) * 0.1, torch.randn(n) - 3
    mb, rb = padded(mean), padded(rho)
    w, e = torch.zeros_like(mb), torch.zeros_like(mb)
    ops.gauss_draw_fwd(mb, rb, w, n, seed=7, stream_id=3, eps_out=e)
    w2 = torch.zeros_like(mb)
    ops.gauss_draw_fwd(mb, rb, w2, n, eps=e)
    assert torch.equal(w[:n], w2[:n])
    gout = padded(torch.randn(n))
    gm1, gr1, gm2, gr2 = (torch.zeros_like(mb) for _ in range(4))
    ops.gauss_draw_bwd(gout, rb, gm1, gr1, n, seed=7, stream_id=3)       # regenerates the forward's noise
    ops.gauss_draw_bwd(gout, rb, gm2, gr2, n, eps=e)
    assert torch.equal(gr1[:n], gr2[:n]) and torch.equal(gm1[:n], gm2[:n])
    # operands that are not 16-byte aligned (per-tensor views into a flat buffer) take the scalar path
    m1, r1, e1 = mb[1:n], rb[1:n], e[1:n]
    wa = torch.zeros(n + 3, device=DEV)
    ops.gauss_draw_fwd(m1, r1, wa[3:3 + n - 1], n - 1, eps=e1)
    ops.gauss_draw_fwd(mb, rb, w2, n, eps=e)
    assert torch.equal(wa[3:3 + n - 1], w2[1:n])
    ga, gb = torch.zeros(n + 1, device=DEV), torch.zeros(n + 1, device=DEV)
    ops.gauss_draw_bwd(gout[1:n], r1, ga[1:n], gb[1:n], n - 1, eps=e1)
    assert torch.equal(ga[1:n], gm2[1:n]) and torch.equal(gb[1:n], gr2[1:n])


# ------------------------------------------------------------------ iVON --
def test_accumulating_unaligned_and_value_only_variants(ops):
    """The instantiations the other tests do not reach (tests/hip_emu/coverage.py): gradients ACCUMULATED into existing
    buffers (draw backward with in-kernel noise, KL, L2), the element-wise fallbacks of the draw kernels for tensors that are
    not 16-byte aligned (views at an odd offset of a parameter buffer), and the value-only forms (no gradient requested)."""
    torch.manual_seed(41)
    n = 10_003
    mean, rho = torch.randn(n, device=DEV) * 0.3, torch.randn(n, device=DEV) - 2.0
    g = torch.randn(n, device=DEV)
    ws = ops.reduce_ws(DEV)
    # draw backward, Philox noise regenerated, overwrite vs accumulate
    gm, gr = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    ops.gauss_draw_bwd(g, rho, gm, gr, n, seed=5, stream_id=6)
    am, ar = torch.full((n,), 0.25, device=DEV), torch.full((n,), -0.5, device=DEV)
    ops.gauss_draw_bwd(g, rho, am, ar, n, seed=5, stream_id=6, accumulate=True)
    assert torch.allclose(am, gm + 0.25, rtol=1e-6, atol=1e-7) and torch.allclose(ar, gr - 0.5, rtol=1e-6, atol=1e-7)
    # the same through views at an odd element offset (4-byte aligned only): forward and backward, supplied and in-kernel noise
    big = lambda: torch.zeros(n + 1, device=DEV)
    bm, br, bw, bg, bgm, bgr = big(), big(), big(), big(), big(), big()
    bm[1:], br[1:], bg[1:] = mean, rho, g
    w_al = torch.empty(n, device=DEV)
    ops.gauss_draw_fwd(mean, rho, w_al, n, seed=5, stream_id=6)
    ops.gauss_draw_fwd(bm[1:], br[1:], bw[1:], n, seed=5, stream_id=6)
    assert torch.allclose(bw[1:], w_al, rtol=1e-6, atol=1e-7)
    eps = torch.randn(n + 1, device=DEV)
    ops.gauss_draw_fwd(bm[1:], br[1:], bw[1:], n, eps=eps[1:])
    assert torch.allclose(bw[1:], mean + torch.nn.functional.softplus(rho) * eps[1:], rtol=2e-6, atol=1e-6)
    ops.gauss_draw_bwd(bg[1:], br[1:], bgm[1:], bgr[1:], n, seed=5, stream_id=6)
    assert torch.allclose(bgm[1:], gm, rtol=1e-6, atol=1e-7) and torch.allclose(bgr[1:], gr, rtol=1e-6, atol=1e-7)
    bgm[1:], bgr[1:] = 0.25, -0.5
    ops.gauss_draw_bwd(bg[1:], br[1:], bgm[1:], bgr[1:], n, seed=5, stream_id=6, accumulate=True)
    assert torch.allclose(bgm[1:], gm + 0.25, rtol=1e-6, atol=1e-7) and torch.allclose(bgr[1:], gr - 0.5, rtol=1e-6, atol=1e-7)
    ops.gauss_draw_bwd(bg[1:], br[1:], bgm[1:], bgr[1:], n, eps=eps[1:], accumulate=True)      # supplied noise, accumulating
    assert torch.isfinite(bgm).all() and torch.isfinite(bgr).all()
    # KL: gradients accumulated; value only
    kl, kg_m, kg_r = torch.zeros(1, device=DEV), torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    ops.gauss_kl(mean, rho, 0.1, 0.7, n, ws, kl_out=kl, gmean=kg_m, grho=kg_r)
    am, ar = torch.full((n,), 0.25, device=DEV), torch.full((n,), -0.5, device=DEV)
    kl2 = torch.zeros(1, device=DEV)
    ops.gauss_kl(mean, rho, 0.1, 0.7, n, ws, kl_out=kl2, gmean=am, grho=ar, accumulate=True)
    assert torch.equal(kl, kl2) and torch.allclose(am, kg_m + 0.25, rtol=1e-6, atol=1e-7) and torch.allclose(ar, kg_r - 0.5, rtol=1e-6, atol=1e-7)
    kl3 = torch.zeros(1, device=DEV)
    ops.gauss_kl(mean, rho, 0.1, 0.7, n, ws, kl_out=kl3)
    assert torch.equal(kl, kl3)
    # L2 of plain parameters (bbb.py:75-76): value only; gradient accumulated
    v1, v2, lg = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV), torch.full((n,), 0.25, device=DEV)
    ops.l2(mean, 0.3, n, ws, val_out=v1)
    ops.l2(mean, 0.3, n, ws, val_out=v2, g=lg, accumulate=True)
    want = 0.5 * 0.3 * (mean.double() ** 2).sum().item()
    assert torch.equal(v1, v2) and abs(v1.item() - want) <= 2e-6 * abs(want)
    assert torch.allclose(lg, 0.25 + 0.3 * mean, rtol=1e-6, atol=1e-7)
    # mixture prior: value only == value of the value + gradient form
    m1, m2, mg = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV), torch.empty(n, device=DEV)
    ops.mixture_nll(mean, 0.5, 1.0, 0.05, n, ws, val_out=m1)
    ops.mixture_nll(mean, 0.5, 1.0, 0.05, n, ws, val_out=m2, gmean=mg)
    assert torch.equal(m1, m2)


def test_ivon_golden_bit_exact(ops, golden):
    g = golden("ivon.npz")
    for ci, (aug, mc, damping, temp) in enumerate(g["cases"]):
        mc = int(mc)
        n_data = 48.0
        init = T(g[f"init_{ci}"])
        d = init.numel()
        mean = padded(init)
        mom = torch.zeros_like(mean)
        prec = torch.full_like(mean, 50.0 / n_data)
        param, dsum = torch.zeros_like(mean), torch.zeros_like(mean)
        eps = T(g[f"eps_{ci}"])
        n_eff = n_data * float(aug)
        for t in range(3):
            for k in range(mc):
                ops.ivon_sample(mean, prec, param, dsum, d, n_eff, first=(k == 0), eps=padded(eps[t * mc + k]))
            # The draw (ivorn.py:108) is anchored on its fp64 evaluation: the sum of the mc draws
            # eps / sqrt(N * clamp(prec, 1e-4)) from the same fp32 inputs.  The reference's fp32 result deviates from
            # that by err_ref (torch's CPU sqrt / reciprocal); ours may deviate by at most twice as much (floor: 1 ulp
            # of the largest draw).  The update below is compared bit for bit.
            want_ds = g[f"delta_sum_{ci}"][t]
            prec64 = prec[:d].cpu().double().clamp(min=1e-4)
            ds64 = sum(eps[t * mc + k].double() / torch.sqrt(n_eff * prec64) for k in range(mc)).numpy()
            err_ref = np.max(np.abs(want_ds.astype(np.float64) - ds64))
            err = np.max(np.abs(dsum[:d].cpu().numpy().astype(np.float64) - ds64))
            assert err <= max(2 * err_ref, 1.2e-7 * np.abs(ds64).max()), (ci, t, err, err_ref)
            last64 = mean[:d].cpu().double().numpy() + (eps[t * mc + mc - 1].double() / torch.sqrt(n_eff * prec64)).numpy()
            err_ref_p = np.max(np.abs(g[f"after_{ci}"][t].astype(np.float64) - last64))
            err_p = np.max(np.abs(param[:d].cpu().numpy().astype(np.float64) - last64))
            assert err_p <= max(2 * err_ref_p, 1.2e-7 * np.abs(last64).max()), (ci, t, err_p, err_ref_p)
            dsum[:d] = T(want_ds).to(DEV)
            ops.ivon_update(mean, mom, prec, dsum, padded(T(g[f"acc_grad_{ci}"][t])), d,
                            lam=float(temp) * 50.0 / n_eff, n_eff=n_eff, mc=mc, beta1=0.9, beta2=0.999, t=t + 1,
                            lr=1e-2, damping=float(damping))
            assert np.array_equal(mean[:d].cpu().numpy(), g[f"means_{ci}"][t]), (ci, t)
            assert np.array_equal(mom[:d].cpu().numpy(), g[f"moms_{ci}"][t]), (ci, t)
            assert np.array_equal(prec[:d].cpu().numpy(), g[f"precs_{ci}"][t]), (ci, t)
        ops.ivon_sample(mean, prec, param, dsum, d, n_eff, first=True, eps=padded(eps[3 * mc]))
        assert np.max(np.abs(param[:d].cpu().numpy() - g[f"eval_sample_{ci}"])) <= 1e-6 * np.abs(eps[3 * mc].numpy()).max()
        ops.ivon_sample(mean, prec, param, dsum, d, n_eff, first=True, deterministic=True)
        assert torch.equal(param[:d], mean[:d])


def test_svgd_step_is_graph_capturable(ops):
    """Every entry point only enqueues on the given stream (no allocation / sync), so a step can be
    captured into a hipGraph and replayed."""
    torch.manual_seed(6)
    m, d = 8, 50021
    P, G = flat_rows(torch.randn(m, d) * 0.05), flat_rows(torch.randn(m, d) * 0.01)
    out_eager, out_graph = torch.zeros_like(G), torch.zeros_like(G)
    ws, ks = ops.svgd_ws(m, DEV), ops.svgd_kstat(m, DEV)
    ops.svgd_step(P, G, out_eager, d, 0.01, 1.0, 5000.0, -1.0, ws, ks)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        ops.svgd_step(P, G, out_graph, d, 0.01, 1.0, 5000.0, -1.0, ws, ks)      # warm-up on the side stream
        s.synchronize()
        with torch.cuda.graph(graph, stream=s):
            ops.svgd_step(P, G, out_graph, d, 0.01, 1.0, 5000.0, -1.0, ws, ks)
    out_graph.zero_()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_graph[:, :d], out_eager[:, :d])


# ------------------------------------------------------------ edge cases --
def test_swag_edge_sizes(ops):
    """Tiny D (scalar tail only), K = 2 / 64 / 256 (maximum), S = 1 / 32 (maximum batch)."""
    torch.manual_seed(8)
    for d, k in [(1, 2), (2, 3), (3, 64), (5, 256), (1027, 256), (4096, 2)]:
        ld = (d + 63) // 64 * 64
        mean, sq = torch.randn(d) * 0.05, None
        devDK = torch.randn(d, k) * 1e-2
        sq = mean ** 2 + torch.rand(d) * 1e-4
        head = k // 3
        ring = torch.zeros(k, ld, device=DEV)
        for c in range(k):
            ring[(head + c) % k, :d] = devDK[:, c].to(DEV)
        ew, ed = torch.randn(k), torch.randn(d)
        out = torch.zeros(ld, device=DEV)
        ops.swag_sample(padded(mean), padded(sq), ring, head, out, d, eps_w=ew.to(DEV), eps_d=padded(ed))
        want = O.swag_sample(mean, sq, devDK, ew, ed)
        assert torch.allclose(out[:d].cpu(), want, rtol=2e-5, atol=2e-6), (d, k)
        for s_count in (1, 32):
            ewb, edb = torch.randn(s_count, k), torch.zeros(s_count, ld, device=DEV)
            edn = torch.randn(s_count, d)
            edb[:, :d] = edn.to(DEV)
            outb = torch.zeros(s_count, ld, device=DEV)
            ops.swag_sample_batched(padded(mean), padded(sq), ring, head, outb, d, eps_w=ewb.to(DEV), eps_d=edb)
            for s in range(s_count):
                want = O.swag_sample(mean, sq, devDK, ewb[s], edn[s])
                assert torch.allclose(outb[s, :d].cpu(), want, rtol=2e-5, atol=2e-6), (d, k, s_count, s)


def test_invalid_arguments_are_rejected(ops):
    from beyond_deep_ensembles_amd.ops import BdeKernelError
    ld = 64
    v = torch.zeros(ld, device=DEV)
    ring = torch.zeros(4, ld, device=DEV)
    with pytest.raises(BdeKernelError):
        ops.swag_sample(v, v, ring, 4, v, 10)                                   # head out of range
    with pytest.raises(BdeKernelError):
        ops.swag_sample(v, v, torch.zeros(300, ld, device=DEV), 0, v, 10)       # K > 256
    with pytest.raises(BdeKernelError):
        ops.swag_sample_batched(v, v, ring, 0, torch.zeros(33, ld, device=DEV), 10)   # S > 32
    with pytest.raises(BdeKernelError):
        ops.swag_update(v[1:], v, v, ring[0], 1, 10)                            # misaligned pointer
    with pytest.raises(BdeKernelError):
        ops.swag_update(v, v, v, ring[0], 0, 10)                                # n must be >= 1
    with pytest.raises(BdeKernelError):
        ops.swag_sample(v.double(), v, ring, 0, v, 10)                          # wrong dtype
    with pytest.raises(BdeKernelError):
        ops.gauss_kl(v, v, 0.0, -1.0, 10, ops.reduce_ws(DEV))                   # prior sigma must be > 0
    P = torch.zeros(8, 66, device=DEV)
    with pytest.raises(BdeKernelError):
        ops.svgd_gram(P, 60, ops.svgd_ws(8, DEV))                               # ld not a multiple of 4 ... (66 % 4 != 0)


def test_randomized_shapes_against_oracle(ops):
    """Seeded fuzz over (M, D, K, S, head, alignment remainders): every op vs the CPU oracle."""
    rng = np.random.default_rng(2024)
    for trial in range(24):
        m = int(rng.integers(1, 17))
        d = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 127, 255, 1000, 4097, 8190, 33333]))
        torch.manual_seed(trial)
        # --- SVGD
        P = torch.randn(m, d) * float(rng.choice([0.01, 0.05, 1.0]))
        if rng.random() < 0.5 and d > 8:                     # shared backbone
            P = P[:1].repeat(m, 1)
            P[:, -max(1, d // 7):] += torch.randn(m, max(1, d // 7)) * 0.02
        G = torch.randn(m, d) * 0.01
        l2, scale, n = float(rng.choice([0.0, 1e-5, 0.01])), float(rng.choice([0.5, 1.0])), float(rng.choice([10.0, 5e4]))
        out, ks = run_svgd(ops, P, G, l2, scale, n)
        phi64 = O.svgd_phi(P.double(), G.double(), l2, scale, n).numpy()
        ref32 = O.svgd_phi(P, G, l2, scale, n).numpy().astype(np.float64)
        err_ref = np.max(np.abs(ref32 - phi64))
        err = np.max(np.abs(-out.numpy() - phi64))
        assert np.isfinite(err) and err <= max(2 * err_ref, 3e-6 * np.max(np.abs(phi64)) + 1e-12), (trial, m, d, err, err_ref)
        # --- SWAG update + sample (+ batched)
        k = int(rng.integers(2, 40))
        head = int(rng.integers(0, k))
        s_count = int(rng.integers(1, 33))
        ld = (d + 63) // 64 * 64
        theta0 = torch.randn(d) * 0.05
        st = O.swag_init(theta0, k)
        mean, sq = padded(st.mean), padded(st.sq_weights)
        ring = torch.zeros(k, ld, device=DEV)
        hd = 0
        for nupd in range(1, int(rng.integers(2, k + 5))):
            theta = theta0 + torch.randn(d) * 1e-2
            st.updates = nupd
            O.swag_moment_update(st, theta)
            ops.swag_update(padded(theta), mean, sq, ring[hd], nupd, d)
            hd = (hd + 1) % k
        assert torch.equal(mean[:d].cpu(), st.mean) and torch.equal(sq[:d].cpu(), st.sq_weights)
        logical = torch.stack([ring[(hd + c) % k, :d].cpu() for c in range(k)], dim=1)
        assert torch.equal(logical, st.deviations), (trial, d, k)
        ew, ed = torch.randn(s_count, k), torch.randn(s_count, d)
        edb = torch.zeros(s_count, ld, device=DEV)
        edb[:, :d] = ed.to(DEV)
        outb = torch.zeros(s_count, ld, device=DEV)
        ops.swag_sample_batched(mean, sq, ring, hd, outb, d, eps_w=ew.to(DEV), eps_d=edb)
        o1 = torch.zeros(ld, device=DEV)
        for s in range(0, s_count, max(1, s_count // 3)):
            want = O.swag_sample(st.mean, st.sq_weights, st.deviations, ew[s], ed[s])
            ops.swag_sample(mean, sq, ring, hd, o1, d, eps_w=ew[s].to(DEV), eps_d=edb[s])
            assert torch.allclose(o1[:d].cpu(), want, rtol=3e-5, atol=3e-6), (trial, d, k, s)
            assert torch.allclose(outb[s, :d].cpu(), want, rtol=3e-5, atol=3e-6), (trial, d, k, s)
        # --- Gaussian KL + draw
        mu, rho, eps = torch.randn(d) * 0.2, torch.randn(d) * 2 - 2, torch.randn(d)
        psig = float(rng.choice([0.1, 1.0, 7.0]))
        ws = ops.reduce_ws(DEV)
        kl, gm, gr = torch.zeros(1, device=DEV), torch.zeros(ld, device=DEV), torch.zeros(ld, device=DEV)
        ops.gauss_kl(padded(mu), padded(rho), 0.1, psig, d, ws, kl_out=kl, gmean=gm, grho=gr, grad_scale=0.5)
        want = O.gauss_kl(mu.double(), rho.double(), 0.1, psig).item()
        assert abs(kl.item() - want) <= 5e-6 * abs(want) + 1e-6, (trial, d)
        wgm, wgr = O.gauss_kl_grads(mu.double(), rho.double(), 0.1, psig)
        s64 = O.gauss_std(rho.double())
        assert np.max(np.abs(gm[:d].cpu().numpy() - 0.5 * wgm.numpy())) <= 3e-6 * (np.max(np.abs(wgm.numpy())) + 1e-3)
        assert np.max(np.abs(gr[:d].cpu().numpy() - 0.5 * wgr.numpy()) / (0.5 * (1 / s64 + s64 / psig ** 2)).numpy()) <= 5e-6
        w = torch.zeros(ld, device=DEV)
        ops.gauss_draw_fwd(padded(mu), padded(rho), w, d, eps=padded(eps))
        assert torch.allclose(w[:d].cpu(), O.gauss_sample(mu, rho, eps), rtol=3e-6, atol=1e-6)


def test_local_reparam_epilogue(ops):
    """out = mean + sqrt(var) * eps and its backward vs torch autograd (bbb_layers.py:70-80 epilogue)."""
    torch.manual_seed(12)
    for n in (3, 64, 1000003):
        mean = torch.randn(n, device=DEV, requires_grad=True)
        var = (torch.rand(n, device=DEV) + 1e-4).requires_grad_()
        eps, gout = torch.randn(n, device=DEV), torch.randn(n, device=DEV)
        (mean + torch.sqrt(var) * eps).backward(gout)
        out, gvar = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
        ops.local_reparam_fwd(mean.detach(), var.detach(), out, n, eps=eps)
        ops.local_reparam_bwd(gout, var.detach(), gvar, n, eps=eps)
        assert torch.allclose(out, (mean + torch.sqrt(var) * eps).detach(), rtol=1e-6, atol=1e-7)
        assert torch.allclose(gvar, var.grad, rtol=2e-6, atol=1e-7)
        # Philox: forward noise regenerated in backward
        out2, gvar2, e2 = torch.empty(n, device=DEV), torch.empty(n, device=DEV), torch.zeros((n + 3) // 4 * 4, device=DEV)
        ops.local_reparam_fwd(mean.detach(), var.detach(), out2, n, seed=3, stream_id=8)
        ops.philox_normal(3, 8, eps_d=e2, d=n)
        assert torch.allclose(out2, (mean + torch.sqrt(var) * e2[:n]).detach(), rtol=1e-6, atol=1e-7)
        ops.local_reparam_bwd(gout, var.detach(), gvar2, n, seed=3, stream_id=8)
        assert torch.allclose(gvar2, gout * e2[:n] / (2 * torch.sqrt(var.detach())), rtol=2e-6, atol=1e-7)


def test_lrt_linear_forward(ops):
    """bde_lrt_linear_fwd (the whole local-reparameterisation forward of BBBLinear, bbb_layers.py:61-80, as one
    fused op: weights streamed once, both products on the MFMA, split-K with a fixed-order finish) against the fp64
    evaluation of those lines; the allowance is twice the deviation of the reference's own fp32 op sequence."""
    import torch.nn.functional as F
    torch.manual_seed(21)
    from oracle import philox as PH
    for b, i, o, bias in [(16, 13, 50, True), (5, 50, 1, True), (16, 2048, 182, True), (128, 300, 70, False),
                          (33, 64, 32, True), (1, 7, 3, True), (96, 1000, 200, True), (64, 4096, 512, False), (70, 129, 33, True),
                          (96, 1000, 1200, True), (20, 640, 2000, True), (128, 2048, 700, True)]:   # the last four: wide path
        x = torch.randn(b, i)
        x[0, : min(i, 3)] = 0.0                                       # exercises the clamp on x^2
        w_mu, w_rho = torch.randn(o, i) * 0.1, torch.randn(o, i) * 1.5 - 3.0
        w_rho[0, : min(i, 4)] = -8.0                                   # ... and on sigma^2
        b_mu, b_rho = (torch.randn(o) * 0.1, torch.randn(o) - 3.0) if bias else (None, None)
        eps = torch.randn(b, o)

        def ref(dt):
            xx, wm, wr = x.to(dt), w_mu.to(dt), w_rho.to(dt)
            mean = F.linear(xx, wm, None if b_mu is None else b_mu.to(dt))
            vb = None if b_rho is None else (F.softplus(b_rho.to(dt)) ** 2).clamp(min=1e-4)
            var = F.linear((xx ** 2).clamp(min=1e-4), (F.softplus(wr) ** 2).clamp(min=1e-4), vb)
            return mean, var
        m64, v64 = ref(torch.float64)
        m32, v32 = ref(torch.float32)
        out64 = m64 + v64.sqrt() * eps.double()
        out32 = m32 + v32.sqrt() * eps
        dev = lambda t: None if t is None else t.to(DEV)
        out, var = torch.empty(b, o, device=DEV), torch.empty(b, o, device=DEV)
        ops.lrt_linear_fwd(dev(x), dev(w_mu), dev(w_rho), dev(b_mu), dev(b_rho), True, out, var, eps=dev(eps))
        tol_v = max(2 * (v32.double() - v64).abs().max().item(), 3e-6 * v64.abs().max().item())
        tol_o = max(2 * (out32.double() - out64).abs().max().item(), 3e-6 * out64.abs().max().item())
        assert (var.cpu().double() - v64).abs().max().item() <= tol_v, (b, i, o)
        assert (out.cpu().double() - out64).abs().max().item() <= tol_o, (b, i, o)
        # deterministic (fixed-order split-K finish), and the in-kernel noise is the Philox stream of element b*O + o
        out2 = torch.empty_like(out)
        ops.lrt_linear_fwd(dev(x), dev(w_mu), dev(w_rho), dev(b_mu), dev(b_rho), True, out2, None, eps=dev(eps))
        assert torch.equal(out, out2)
        ops.lrt_linear_fwd(dev(x), dev(w_mu), dev(w_rho), dev(b_mu), dev(b_rho), True, out2, None, seed=9, stream_id=4)
        z = torch.from_numpy(PH.normals(9, 4, b * o)).view(b, o)
        assert (out2.cpu().double() - (m64 + v64.sqrt() * z)).abs().max().item() <= tol_o + 5e-6 * v64.sqrt().max().item()
        # a strided input (a column slice of a wider activation matrix)
        wide = torch.randn(b, i + 5, device=DEV)
        wide[:, :i] = dev(x)
        ops.lrt_linear_fwd(wide[:, :i], dev(w_mu), dev(w_rho), dev(b_mu), dev(b_rho), True, out2, None, eps=dev(eps))
        assert (out2 - out).abs().max().item() <= tol_o
    assert not ops.lrt_linear_supported(129, 10, 10) and ops.lrt_linear_supported(128, 10, 10)


CONV_CASES = [
    # (N, C, H, W, O, K, stride, padding, bias)            the CIFAR ResNet-20 layer shapes of BASELINE configs[1] ...
    (8, 3, 32, 32, 16, 3, (1, 1), (1, 1), True), (8, 16, 32, 32, 16, 3, (1, 1), (1, 1), True),
    (8, 16, 32, 32, 32, 3, (2, 2), (1, 1), True), (8, 32, 16, 16, 32, 3, (1, 1), (1, 1), False),
    (8, 32, 16, 16, 64, 3, (2, 2), (1, 1), True), (9, 64, 8, 8, 64, 3, (1, 1), (1, 1), True),
    (4, 16, 32, 32, 32, 1, (2, 2), (0, 0), False),                                       # a 1x1 stride-2 shortcut
    # ... and ragged ones: odd channel counts, rectangular images, no / wide padding, per-axis strides, 5x5 and 7x7 kernels
    (3, 5, 9, 11, 7, 3, (1, 1), (0, 0), True), (2, 7, 13, 6, 33, 3, (2, 1), (1, 2), True), (5, 4, 12, 12, 20, 5, (1, 1), (2, 2), True),
    (2, 3, 30, 30, 64, 7, (2, 2), (3, 3), False), (1, 130, 7, 7, 40, 3, (1, 1), (1, 1), True), (2, 64, 14, 14, 64, 1, (1, 1), (0, 0), True),
    (3, 20, 28, 28, 16, 3, (1, 1), (1, 1), True),
    (1, 256, 6, 56, 64, 1, (1, 1), (0, 0), False),          # wide 1x1 on a wide image: the weight gradient runs with fewer column tiles
    # few output channels / single images: the remaining (channel tile, pixel tiles per wave) instantiations of the kernel
    (1, 3, 48, 48, 4, 1, (2, 2), (0, 0), True), (1, 3, 32, 32, 4, 3, (1, 1), (1, 1), False), (1, 3, 48, 48, 40, 1, (1, 1), (0, 0), True),
    (8, 3, 32, 32, 4, 1, (1, 1), (0, 0), False), (1, 3, 16, 16, 4, 3, (1, 1), (1, 1), True),
    # 7 and 8 column tiles of the weight gradient per workgroup (conv_lrt_wgrad_kernel<16, 7> / <16, 8>)
    (2, 12, 8, 8, 8, 3, (1, 1), (1, 1), True), (2, 14, 8, 8, 6, 3, (1, 1), (1, 1), False),
]


@unverified("conv_lrt")
def test_conv_lrt_forward(ops):
    """bde_conv_lrt_fwd -- BBBConv2d's two convolutions (bbb_layers.py:146-147) as one dual-accumulator implicit GEMM
    with the sampling epilogue (148-154) -- against the fp64 evaluation of those lines; the allowance is twice the
    deviation of the reference's own fp32 op sequence (torch CPU conv2d)."""
    import torch.nn.functional as F
    from oracle import philox as PH
    torch.manual_seed(33)
    for n, c, h, w, o, k, stride, padding, bias in CONV_CASES:
        x = torch.randn(n, c, h, w)
        x[0, 0, :2] = 0.0                                               # exercises the clamp on x^2
        w_mu, w_rho = torch.randn(o, c, k, k) * 0.1, torch.randn(o, c, k, k) * 1.5 - 3.0
        w_rho[0, 0] = -8.0                                              # ... and on sigma^2
        b_mu, b_rho = (torch.randn(o) * 0.1, torch.randn(o) - 3.0) if bias else (None, None)

        def ref(dt):
            xx = x.to(dt)
            mean = F.conv2d(xx, w_mu.to(dt), None if b_mu is None else b_mu.to(dt), stride=stride, padding=padding)
            var = F.conv2d((xx ** 2).clamp(min=1e-4), (F.softplus(w_rho.to(dt)) ** 2).clamp(min=1e-4),
                           None if b_rho is None else F.softplus(b_rho.to(dt)) ** 2, stride=stride, padding=padding)
            return mean, var
        m64, v64 = ref(torch.float64)
        m32, v32 = ref(torch.float32)
        eps = torch.randn(m32.shape)
        out64, out32 = m64 + v64.sqrt() * eps.double(), m32 + v32.sqrt() * eps
        dev = lambda t: None if t is None else t.to(DEV).contiguous()
        xd, wm = dev(x), dev(w_mu)
        wbuf = ops.conv_lrt_wbuf(w_mu.shape, DEV)
        ops.conv_lrt_prep(wm, dev(w_rho), wbuf, dev(b_rho))
        bvar = bias
        assert ops.conv_lrt_supported(x.shape, w_mu.shape, stride, padding), (n, c, h, w, o, k)
        out, var = torch.full(m32.shape, 9.0, device=DEV), torch.full(m32.shape, 9.0, device=DEV)
        ops.conv_lrt_fwd(xd, wbuf, w_mu.shape, dev(b_mu), bvar, stride, padding, out, var, eps=dev(eps))
        # fp32 accumulation over c*k*k products in one chain: the relative allowance grows with the square root of its length
        # beyond 1024 (all of CONV_CASES are below; tests/hip_emu/sweep_conv.py also runs 512-channel layers)
        rel = 3e-6 * max(1.0, (c * k * k / 1024.0) ** 0.5)
        tol_v = max(2 * (v32.double() - v64).abs().max().item(), rel * v64.abs().max().item())
        tol_o = max(2 * (out32.double() - out64).abs().max().item(), rel * out64.abs().max().item())
        case = (n, c, h, w, o, k, stride, padding)
        assert (var.cpu().double() - v64).abs().max().item() <= tol_v, case
        assert (out.cpu().double() - out64).abs().max().item() <= tol_o, case
        # deterministic, and the in-kernel noise is the Philox stream of the flat NCHW output element (the numbering of
        # bde_local_reparam_fwd, whose backward regenerates it)
        out2, var2 = torch.empty_like(out), torch.empty_like(var)
        ops.conv_lrt_fwd(xd, wbuf, w_mu.shape, dev(b_mu), bvar, stride, padding, out2, var2, eps=dev(eps))
        assert torch.equal(out, out2) and torch.equal(var, var2)
        ops.conv_lrt_fwd(xd, wbuf, w_mu.shape, dev(b_mu), bvar, stride, padding, out2, var2, seed=9, stream_id=4)
        z = torch.from_numpy(PH.normals(9, 4, out2.numel())).view(out2.shape)
        assert (out2.cpu().double() - (m64 + v64.sqrt() * z)).abs().max().item() <= tol_o + 5e-6 * v64.sqrt().max().item(), case
        assert torch.equal(var, var2)


@unverified("conv_lrt")
def test_conv_lrt_backward(ops):
    """bde_local_reparam_bwd + bde_conv_lrt_bwd_data + bde_conv_lrt_bwd_weight (the autograd graph of
    bbb_layers.py:146-154: two transposed and two weight-gradient convolutions + the element-wise chain) against fp64
    autograd of those lines; the allowance is twice the deviation of fp32 autograd of the same lines (torch CPU)."""
    import torch.nn.functional as F
    torch.manual_seed(35)
    for n, c, h, w, o, k, stride, padding, bias in CONV_CASES:
        x = torch.randn(n, c, h, w)
        x[0, 0, :2] = 0.0
        x[0, 0, 2, : min(w, 3)] = 5e-3                                  # x^2 below the clamp but not zero
        w_mu, w_rho = torch.randn(o, c, k, k) * 0.1, torch.randn(o, c, k, k) * 1.5 - 3.0
        w_rho[0, 0] = -8.0
        b_mu, b_rho = (torch.randn(o) * 0.1, torch.randn(o) - 3.0) if bias else (None, None)
        eps, gout = None, None

        def run(dt):
            leaves = [t.to(dt).clone().requires_grad_(True) for t in (x, w_mu, w_rho)]
            xx, wm, wr = leaves
            bm = None if b_mu is None else b_mu.to(dt)
            br = None if b_rho is None else b_rho.to(dt)
            mean = F.conv2d(xx, wm, bm, stride=stride, padding=padding)
            var = F.conv2d((xx ** 2).clamp(min=1e-4), (F.softplus(wr) ** 2).clamp(min=1e-4),
                           None if br is None else F.softplus(br) ** 2, stride=stride, padding=padding)
            nonlocal eps, gout
            if eps is None:
                eps, gout = torch.randn(mean.shape, dtype=torch.float64), torch.randn(mean.shape, dtype=torch.float64)
            outv = mean + var.sqrt() * eps.to(dt)
            return [t.detach() for t in torch.autograd.grad(outv, leaves, gout.to(dt))] + [var.detach()]
        g64 = run(torch.float64)
        g32 = run(torch.float32)
        dev = lambda t: None if t is None else t.to(DEV).float().contiguous()
        xd, wm, wr = dev(x), dev(w_mu), dev(w_rho)
        wbuf = ops.conv_lrt_wbuf(w_mu.shape, DEV)
        ops.conv_lrt_prep(wm, wr, wbuf, stride=stride, padding=padding)   # with the stride: + the per-phase input-gradient matrices
        var = dev(g32[3])
        gd, ed = dev(gout), dev(eps)
        gvar = torch.empty_like(gd)
        ops.local_reparam_bwd(gd.view(-1), var.view(-1), gvar.view(-1), gd.numel(), eps=ed.view(-1))
        gx = torch.full_like(xd, 9.0)
        ops.conv_lrt_bwd_data(gd, gvar, wbuf, w_mu.shape, xd, gx, stride, padding, phases=True)   # one launch per phase if strided
        gx_dilated = torch.full_like(xd, 9.0)                              # ... and the single launch over the zero-dilated gradient
        ops.conv_lrt_bwd_data(gd, gvar, wbuf, w_mu.shape, xd, gx_dilated, stride, padding)
        gwm, gwr = torch.full_like(wm, 9.0), torch.full_like(wr, 9.0)
        ops.conv_lrt_bwd_weight(xd, gd, gvar, wr, gwm, gwr, stride, padding)
        case = (n, c, h, w, o, k, stride, padding)
        ho, wo = g32[3].shape[2:]
        chain = {"g_x": o * k * k, "g_wmu": n * ho * wo, "g_wrho": n * ho * wo}         # products summed per output element
        for name, got, i in (("g_x", gx, 0), ("g_x", gx_dilated, 0), ("g_wmu", gwm, 1), ("g_wrho", gwr, 2)):
            rel = 3e-6 * max(1.0, (chain[name] / 1024.0) ** 0.5)
            tol = max(2 * (g32[i].double() - g64[i]).abs().max().item(), rel * g64[i].abs().max().item())
            assert (got.cpu().double() - g64[i]).abs().max().item() <= tol, (name, case)
        gwm2, gwr2, gx2 = torch.empty_like(gwm), torch.empty_like(gwr), torch.empty_like(gx)     # deterministic
        ops.conv_lrt_bwd_weight(xd, gd, gvar, wr, gwm2, gwr2, stride, padding)
        ops.conv_lrt_bwd_data(gd, gvar, wbuf, w_mu.shape, xd, gx2, stride, padding, phases=True)
        assert torch.equal(gwm, gwm2) and torch.equal(gwr, gwr2) and torch.equal(gx, gx2), case


@unverified("conv_lrt")
def test_r5_conv_gvar_and_bias_gradients_in_one_pass(ops):
    """bde_conv_lrt_gvar_bias: g_var = g eps / (2 sqrt(var)) over the layer output -- bit for bit what bde_local_reparam_bwd
    writes (same arithmetic, same Philox stream) -- and the two bias gradients of bbb_layers.py:146-147 from the same pass,
    against fp64 (channel sums of g and g_var, the rho chain rule of the UNclamped bias variance), for plane sizes that are
    and are not multiples of 4, supplied and in-kernel noise, with and without a bias."""
    import torch.nn.functional as F
    torch.manual_seed(52)
    for n, o, ho, wo in [(3, 5, 4, 4), (2, 7, 3, 5), (8, 16, 32, 32), (5, 33, 8, 8), (1, 3, 1, 1), (70, 4, 6, 6)]:
        g = torch.randn(n, o, ho, wo)
        var = torch.rand(n, o, ho, wo) + 0.05
        b_rho = torch.randn(o) - 2.0
        dev = lambda t: t.to(DEV).contiguous()
        gd, vd = dev(g), dev(var)
        for supplied in (True, False):
            eps = torch.randn(n, o, ho, wo) if supplied else None
            want = torch.empty_like(gd)
            ops.local_reparam_bwd(gd.view(-1), vd.view(-1), want.view(-1), gd.numel(), eps=None if eps is None else dev(eps).view(-1),
                                  seed=11, stream_id=3)
            for bias in (True, False):
                gvar = torch.full_like(gd, 9.0)
                g_bmu, g_brho = (torch.full((o,), 9.0, device=DEV), torch.full((o,), 9.0, device=DEV)) if bias else (None, None)
                ops.conv_lrt_gvar_bias(gd, vd, gvar, eps=None if eps is None else dev(eps), seed=11, stream_id=3,
                                       b_rho=dev(b_rho) if bias else None, g_bmu=g_bmu, g_brho=g_brho)
                assert torch.equal(gvar, want), (n, o, ho, wo, supplied, bias)
                if bias:
                    gv64 = want.cpu().double()
                    sp = F.softplus(b_rho.double())
                    want_bmu = g.double().sum(dim=(0, 2, 3))
                    want_brho = gv64.sum(dim=(0, 2, 3)) * 2.0 * sp * torch.sigmoid(b_rho.double())
                    scale = max(1.0, float(n * ho * wo) ** 0.5)
                    assert (g_bmu.cpu().double() - want_bmu).abs().max().item() <= 2e-6 * scale * max(1.0, g.abs().max().item())
                    assert (g_brho.cpu().double() - want_brho).abs().max().item() <= 2e-6 * scale * max(1.0, gv64.abs().max().item())
                    again_m, again_r = torch.empty_like(g_bmu), torch.empty_like(g_brho)        # deterministic
                    ops.conv_lrt_gvar_bias(gd, vd, gvar, eps=None if eps is None else dev(eps), seed=11, stream_id=3,
                                           b_rho=dev(b_rho), g_bmu=again_m, g_brho=again_r)
                    assert torch.equal(again_m, g_bmu) and torch.equal(again_r, g_brho)


TILING_MIN_PAIRS = 20
TILING_CASES = [(2, 5, 9, 11, 7, 3, (1, 1), (0, 0)), (2, 16, 12, 12, 32, 3, (2, 2), (1, 1)), (3, 20, 10, 10, 16, 3, (1, 1), (1, 1)),
                (2, 40, 6, 6, 40, 1, (1, 1), (0, 0)), (1, 8, 9, 9, 8, 3, (3, 3), (2, 2))]


@unverified("conv_lrt")
def test_r5_conv_every_candidate_tiling_computes_the_same_layer(ops):
    """The tuning hooks of the fused convolution (bde_conv_lrt_pass_geos / _candidates / _set_tiling and the weight-gradient
    pair): EVERY tiling the planners enumerate for a layer -- not only the one their hand-set score picks -- is pinned in
    turn and must compute the layer (forward, dilated and per-phase input gradient, weight gradient) to the tolerance of the
    fp64 anchor; a tiling that is not a candidate is refused; removing the pin restores the planner's choice bit for bit.
    This is what tools/conv_autotune.py relies on when it times the candidates on the device and pins the winners."""
    import torch.nn.functional as F
    from beyond_deep_ensembles_amd.ops import BdeKernelError
    torch.manual_seed(41)
    total = 0
    for n, c, h, w, o, k, stride, padding in TILING_CASES:
        x = torch.randn(n, c, h, w)
        w_mu, w_rho = torch.randn(o, c, k, k) * 0.1, torch.randn(o, c, k, k) * 1.5 - 3.0
        xs, ws = tuple(x.shape), tuple(w_mu.shape)

        def run(dt):
            leaves = [t.to(dt).clone().requires_grad_(True) for t in (x, w_mu, w_rho)]
            xx, wm, wr = leaves
            mean = F.conv2d(xx, wm, None, stride=stride, padding=padding)
            var = F.conv2d((xx ** 2).clamp(min=1e-4), (F.softplus(wr) ** 2).clamp(min=1e-4), None, stride=stride, padding=padding)
            return mean, var, leaves
        m64, v64, l64 = run(torch.float64)
        eps, gout = torch.randn(m64.shape, dtype=torch.float64), torch.randn(m64.shape, dtype=torch.float64)
        g64 = [t.detach() for t in torch.autograd.grad(m64 + v64.sqrt() * eps, l64, gout)]
        out64 = (m64 + v64.sqrt() * eps).detach()
        dev = lambda t: t.to(DEV).float().contiguous()
        xd, wm, wr, ed, gd = dev(x), dev(w_mu), dev(w_rho), dev(eps), dev(gout)
        wbuf = ops.conv_lrt_wbuf(ws, DEV)
        ops.conv_lrt_prep(wm, wr, wbuf, stride=stride, padding=padding)
        shape = tuple(m64.shape)

        def forward():
            out, var = torch.full(shape, 9.0, device=DEV), torch.full(shape, 9.0, device=DEV)
            ops.conv_lrt_fwd(xd, wbuf, ws, None, False, stride, padding, out, var, eps=ed)
            return out, var
        out0, var0 = forward()
        gvar = torch.empty_like(gd)
        ops.local_reparam_bwd(gd.view(-1), var0.view(-1), gvar.view(-1), gd.numel(), eps=ed.view(-1))

        def bwd_data(phases):
            gx = torch.full_like(xd, 9.0)
            ops.conv_lrt_bwd_data(gd, gvar, wbuf, ws, xd, gx, stride, padding, phases=phases)
            return gx

        def bwd_weight():
            gwm, gwr = torch.full_like(wm, 9.0), torch.full_like(wr, 9.0)
            ops.conv_lrt_bwd_weight(xd, gd, gvar, wr, gwm, gwr, stride, padding)
            return gwm, gwr

        def close(got, want64, chain):
            rel = 2e-5 * max(1.0, (chain / 1024.0) ** 0.5)
            return (got.cpu().double() - want64).abs().max().item() <= rel * want64.abs().max().item()
        passes = [(0, forward, lambda r: close(r[0], out64, c * k * k) and close(r[1], v64.detach(), c * k * k)),
                  (1, lambda: bwd_data(False), lambda r: close(r, g64[0], o * k * k)),
                  (2, lambda: bwd_data(True), lambda r: close(r, g64[0], o * k * k))]
        for which, fn, ok in passes:
            base = fn()
            assert ok(base), (which, xs, ws)
            for geo in ops.conv_lrt_pass_geos(which, xs, ws, stride, padding):
                cands, chosen = ops.conv_lrt_candidates(geo)
                assert cands and 0 <= chosen < len(cands), geo
                for cand in cands:
                    ops.conv_lrt_set_tiling(geo, cand)
                    assert ops.conv_lrt_candidates(geo)[1] == cands.index(cand)
                    assert ok(fn()), (which, geo, cand)
                    total += 1
                with pytest.raises(BdeKernelError):
                    ops.conv_lrt_set_tiling(geo, (3, 1, 1, 1))                  # 3 k-splits: never a candidate
                ops.conv_lrt_set_tiling(geo, None)
                assert ops.conv_lrt_candidates(geo)[1] == chosen
            again = fn()
            for a, b in zip(base if isinstance(base, tuple) else (base,), again if isinstance(again, tuple) else (again,)):
                assert torch.equal(a, b), (which, xs, ws)
        ho, wo = shape[2:]
        base = bwd_weight()
        cands, chosen = ops.conv_lrt_wgrad_candidates(xs, ws, stride, padding)
        assert cands and 0 <= chosen < len(cands)
        for cand in cands:
            ops.conv_lrt_wgrad_set_tiling(xs, ws, stride, padding, cand)
            assert ops.conv_lrt_wgrad_candidates(xs, ws, stride, padding)[1] == cands.index(cand)
            gwm, gwr = bwd_weight()
            assert close(gwm, g64[1], n * ho * wo) and close(gwr, g64[2], n * ho * wo), (xs, ws, cand)
            total += 1
        with pytest.raises(BdeKernelError):
            ops.conv_lrt_wgrad_set_tiling(xs, ws, stride, padding, (99, 1, 1, 1))
        ops.conv_lrt_wgrad_set_tiling(xs, ws, stride, padding, None)
        again = bwd_weight()
        assert torch.equal(base[0], again[0]) and torch.equal(base[1], again[1])
    assert total > TILING_MIN_PAIRS * len(TILING_CASES)                  # dozens of (pass, tiling) pairs per layer


def test_lrt_linear_backward(ops):
    """bde_lrt_linear_bwd (the autograd graph of bbb_layers.py:61-80 in three or four launches) against fp64 autograd
    over those lines; the allowance is twice the deviation of fp32 autograd over the same lines (CPU)."""
    import torch.nn.functional as F
    from oracle import philox as PH
    torch.manual_seed(22)
    dev = lambda t: None if t is None else t.to(DEV)
    for b, i, o, bias, clamp_bias in [(16, 13, 50, True, True), (5, 50, 1, True, True), (16, 2048, 182, True, True),
                                      (128, 300, 70, False, True), (33, 64, 32, True, False), (1, 7, 3, True, True),
                                      (96, 1000, 200, True, True), (64, 2100, 520, True, True), (70, 129, 33, True, True),
                                      # tile-aligned wide layers: the mask-free variants of the two gradient kernels
                                      (64, 1056, 1024, True, True), (32, 2048, 544, False, True),
                                      # wide layers whose sizes are no multiples of 4, batch <= 64: lrt_bwd_x_kernel<1 / 2>
                                      (32, 1030, 1027, True, True), (64, 1027, 1030, False, True),
                                      # few outputs x very many inputs: ONE slice over O, the input gradient written directly
                                      # (DIRECT) -- float4 kernel mask-free / masked at 1 and 2 batch tiles, the row kernel at 1, 2, 4
                                      (32, 16384, 96, True, True), (64, 16384, 96, False, True), (32, 16388, 68, True, True),
                                      (64, 16388, 68, True, True), (16, 12004, 90, True, True), (40, 12004, 90, False, True),
                                      (96, 12004, 90, True, True)]:
        x = torch.randn(b, i)
        x[0, : min(i, 3)] = 0.0                                       # x^2 below the clamp: no gradient through it
        w_mu, w_rho = torch.randn(o, i) * 0.1, torch.randn(o, i) * 1.5 - 3.0
        w_rho[0, : min(i, 4)] = -8.0                                   # sigma^2 below the clamp
        b_rho = torch.randn(o) - 3.0 if bias else None
        if bias:
            b_rho[0] = -8.0
        eps, g = torch.randn(b, o), torch.randn(b, o)

        def ref(dt):
            leaves = [t.to(dt).requires_grad_(True) for t in (x, w_mu, w_rho)]
            xx, wm, wr = leaves
            bm = br = vb = None
            if bias:
                bm, br = torch.zeros(o, dtype=dt, requires_grad=True), b_rho.to(dt).requires_grad_(True)
                vb = F.softplus(br) ** 2
                if clamp_bias:
                    vb = vb.clamp(min=1e-4)
                leaves += [bm, br]
            var = F.linear((xx ** 2).clamp(min=1e-4), (F.softplus(wr) ** 2).clamp(min=1e-4), vb)
            out = F.linear(xx, wm, bm) + var.sqrt() * eps.to(dt)
            return var.detach(), [t.double() for t in torch.autograd.grad(out, leaves, grad_outputs=g.to(dt))]
        v64, g64 = ref(torch.float64)
        v32, g32 = ref(torch.float32)
        var = dev(v32)
        outs = [torch.full((b, i), float("nan"), device=DEV), torch.full((o, i), float("nan"), device=DEV),
                torch.full((o, i), float("nan"), device=DEV)]
        outs += [torch.full((o,), float("nan"), device=DEV), torch.full((o,), float("nan"), device=DEV)] if bias else [None, None]
        ops.lrt_linear_bwd(dev(x), dev(w_mu), dev(w_rho), dev(b_rho), clamp_bias, dev(g), var, *outs, eps=dev(eps))
        names = ["g_x", "g_wmu", "g_wrho", "g_bmu", "g_brho"]
        for name, ours, r64, r32 in zip(names, outs, g64, g32):
            tol = max(2 * (r32 - r64).abs().max().item(), 3e-6 * r64.abs().max().item())
            assert (ours.cpu().double() - r64).abs().max().item() <= tol, (name, b, i, o)
        # the clamp masks: exactly zero where the reference's clamp blocks the gradient
        assert (outs[2][0, : min(i, 4)] == 0).all()
        if bias and clamp_bias:
            assert outs[4][0].item() == 0.0
        # bit-reproducible; g_x optional; in-kernel noise = the Philox stream the forward used
        again = [torch.empty_like(t) if t is not None else None for t in outs]
        ops.lrt_linear_bwd(dev(x), dev(w_mu), dev(w_rho), dev(b_rho), clamp_bias, dev(g), var, *again, eps=dev(eps))
        assert all(a is None or torch.equal(a, t) for a, t in zip(again, outs))
        again[0] = None
        ops.lrt_linear_bwd(dev(x), dev(w_mu), dev(w_rho), dev(b_rho), clamp_bias, dev(g), var, *again, eps=dev(eps))
        assert torch.equal(again[1], outs[1]) and torch.equal(again[2], outs[2])
        z = torch.from_numpy(PH.normals(9, 4, b * o)).view(b, o).float()
        with_z = [torch.empty_like(t) if t is not None else None for t in outs]
        ops.lrt_linear_bwd(dev(x), dev(w_mu), dev(w_rho), dev(b_rho), clamp_bias, dev(g), var, *with_z, eps=dev(z))
        philox = [torch.empty_like(t) if t is not None else None for t in outs]
        ops.lrt_linear_bwd(dev(x), dev(w_mu), dev(w_rho), dev(b_rho), clamp_bias, dev(g), var, *philox, seed=9, stream_id=4)
        for name, a, c in zip(names, philox, with_z):
            if a is not None:
                assert (a - c).abs().max().item() <= 2e-5 * max(c.abs().max().item(), 1e-6), name
        # strided input rows
        wide = torch.randn(b, i + 5, device=DEV)
        wide[:, :i] = dev(x)
        strided = [torch.empty_like(t) if t is not None else None for t in outs]
        ops.lrt_linear_bwd(wide[:, :i], dev(w_mu), dev(w_rho), dev(b_rho), clamp_bias, dev(g), var, *strided, eps=dev(eps))
        assert all(a is None or torch.equal(a, t) for a, t in zip(strided, outs))


def test_var_operand_kernels(ops):
    """bde_var_operand_fwd/bwd (the operands of the variance product, bbb_layers.py:66-67,71,150-153) against the
    reference's op sequence and its autograd, incl. the clamp boundaries, ragged sizes and the mode without clamp."""
    import torch.nn.functional as F
    torch.manual_seed(23)
    for n in (1, 3, 4, 5, 1000, 4099, 1 << 20):
        v = torch.randn(n) * 2 - 1
        v[0] = 0.0                                   # x^2 below the clamp; softplus(0)^2 = 0.48 above it
        if n > 2:
            v[1], v[2] = -9.0, 0.01                  # softplus(-9)^2 = 1.5e-8 below the clamp; x = 0.01: x^2 = 1e-4 (boundary)
        g = torch.randn(n)
        for mode in (0, 1, 2):
            leaf = v.clone().double().requires_grad_(True)
            if mode == 0:
                ref = (leaf ** 2).clamp(min=1e-4)
            else:
                ref = F.softplus(leaf) ** 2
                if mode == 1:
                    ref = ref.clamp(min=1e-4)
            gref = torch.autograd.grad(ref, leaf, grad_outputs=g.double())[0]
            out, gv = torch.full((n,), float("nan"), device=DEV), torch.full((n,), float("nan"), device=DEV)
            ops.var_operand_fwd(v.to(DEV), mode, out)
            ops.var_operand_bwd(g.to(DEV), v.to(DEV), mode, gv)
            assert (out.cpu().double() - ref.detach()).abs().max().item() <= 1e-6 * max(1.0, ref.abs().max().item()), (n, mode)
            # the masks of the clamp are decided in fp32 on the device: compare away from the boundary |value - 1e-4| tiny
            val32 = (v ** 2) if mode == 0 else F.softplus(v) ** 2
            safe = ((val32 - 1e-4).abs() > 1e-9) | torch.tensor(mode == 2)
            err = (gv.cpu().double() - gref).abs()
            assert err[safe].max().item() <= 2e-6 * max(1.0, gref.abs().max().item()), (n, mode)
            if mode == 1 and n > 2:
                assert gv[1].item() == 0.0 and out[1].item() == pytest.approx(1e-4)
            if mode == 2 and n > 2:
                assert gv[1].item() != 0.0


def test_lrt_sigma_cache_is_bit_identical(ops):
    """Wide layers: sigma^2 = clamp(softplus(rho)^2, 1e-4) and its rho-derivative computed ONCE per weight version
    (bde_lrt_sigma_cache) and read by the fused forward / backward instead of evaluating softplus / sigmoid per weight per
    pass.  Same expressions, so outputs and all five gradients equal the on-the-fly kernels bit for bit; narrow layers
    ignore the cache."""
    torch.manual_seed(29)
    for b, i, o in [(64, 1024, 1100), (128, 2048, 700), (20, 640, 2000), (33, 4096, 512), (16, 256, 64),
                    # every remaining (batch tiles, DIRECT, FULL) form of the two input-gradient kernels, cached vs on the fly
                    (32, 2048, 544), (64, 1056, 1024), (32, 1030, 1027), (64, 1027, 1030), (32, 16384, 96), (64, 16384, 96),
                    (32, 16388, 68), (64, 16388, 68), (16, 12004, 90), (40, 12004, 90), (96, 12004, 90)]:
        wide = ops.lrt_sigma_cache_wanted(i, o)
        assert wide == (i * o >= (1 << 20) and i % 4 == 0 and i >= 512)   # (the backward reads a cache it is handed for any wide layer)
        x = torch.randn(b, i, device=DEV)
        x[0, :3] = 0.0
        w_mu, w_rho = torch.randn(o, i, device=DEV) * 0.1, torch.randn(o, i, device=DEV) * 1.5 - 3.0
        w_rho[0, :4] = -8.0                                            # below the clamp: the mask of the chain rule
        b_mu, b_rho = torch.randn(o, device=DEV) * 0.1, torch.randn(o, device=DEV) - 3.0
        s2, ds2 = torch.empty_like(w_rho), torch.empty_like(w_rho)
        ops.lrt_sigma_cache(w_rho, s2, ds2)
        sp = torch.nn.functional.softplus(w_rho.double())
        assert (s2.double() - (sp ** 2).clamp(min=1e-4)).abs().max().item() <= 3e-7 * float((sp ** 2).max())
        want_ds2 = (sp ** 2 >= 1e-4).double() * 2 * sp * torch.sigmoid(w_rho.double())
        assert (ds2.double() - want_ds2).abs().max().item() <= 1e-6 * float(want_ds2.abs().max()) + 1e-9
        outs = []
        for cached in (False, True):
            out, var = torch.empty(b, o, device=DEV), torch.empty(b, o, device=DEV)
            ops.lrt_linear_fwd(x, w_mu, w_rho, b_mu, b_rho, True, out, var, seed=5, stream_id=2, w_s2=s2 if cached else None)
            g = torch.randn(b, o, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
            gs = [torch.empty(b, i, device=DEV), torch.empty_like(w_mu), torch.empty_like(w_rho), torch.empty_like(b_mu),
                  torch.empty_like(b_rho)]
            ops.lrt_linear_bwd(x, w_mu, w_rho, b_rho, True, g, var, *gs, seed=5, stream_id=2,
                               w_s2=s2 if cached else None, w_ds2=ds2 if cached else None)
            outs.append([out, var] + gs)
        torch.cuda.synchronize()
        for name, a, c in zip(("out", "var", "g_x", "g_wmu", "g_wrho", "g_bmu", "g_brho"), *outs):
            assert torch.equal(a, c), (b, i, o, name)


def test_lrt_linear_random_shapes(ops):
    """Forward + backward of the fused BBBLinear ops at seeded random shapes (ragged B / I / O, both the row-per-lane
    and the wide LDS-staged forward kernels, split and unsplit input-gradient reductions) against fp64 autograd over
    bbb_layers.py:70-80; allowance = twice the deviation of fp32 autograd over the same lines."""
    import torch.nn.functional as F
    rs = np.random.RandomState(77)
    shapes = [(int(rs.randint(1, 129)), int(rs.randint(1, 400)), int(rs.randint(1, 300))) for _ in range(10)]
    shapes += [(int(rs.randint(1, 129)), 4 * int(rs.randint(130, 600)), int(rs.randint(450, 900))) for _ in range(4)]   # wide
    torch.manual_seed(78)
    dev = lambda t: None if t is None else t.to(DEV)
    for b, i, o in shapes:
        bias = bool(rs.randint(0, 2))
        x, eps, g = torch.randn(b, i), torch.randn(b, o), torch.randn(b, o)
        w_mu, w_rho = torch.randn(o, i) * 0.1, torch.randn(o, i) * 1.5 - 3.0
        b_mu, b_rho = (torch.randn(o) * 0.1, torch.randn(o) - 3.0) if bias else (None, None)

        def ref(dt):
            leaves = [t.to(dt).requires_grad_(True) for t in (x, w_mu, w_rho)]
            xx, wm, wr = leaves
            bm = br = vb = None
            if bias:
                bm, br = b_mu.to(dt).requires_grad_(True), b_rho.to(dt).requires_grad_(True)
                vb = (F.softplus(br) ** 2).clamp(min=1e-4)
                leaves += [bm, br]
            var = F.linear((xx ** 2).clamp(min=1e-4), (F.softplus(wr) ** 2).clamp(min=1e-4), vb)
            out = F.linear(xx, wm, bm) + var.sqrt() * eps.to(dt)
            return out.detach().double(), var.detach(), [t.double() for t in torch.autograd.grad(out, leaves, grad_outputs=g.to(dt))]
        o64, v64, g64 = ref(torch.float64)
        o32, v32, g32 = ref(torch.float32)
        out, var = torch.empty(b, o, device=DEV), torch.empty(b, o, device=DEV)
        ops.lrt_linear_fwd(dev(x), dev(w_mu), dev(w_rho), dev(b_mu), dev(b_rho), True, out, var, eps=dev(eps))
        tol = max(2 * (o32 - o64).abs().max().item(), 3e-6 * o64.abs().max().item())
        assert (out.cpu().double() - o64).abs().max().item() <= tol, ("out", b, i, o)
        outs = [torch.empty(b, i, device=DEV), torch.empty(o, i, device=DEV), torch.empty(o, i, device=DEV)]
        outs += [torch.empty(o, device=DEV), torch.empty(o, device=DEV)] if bias else [None, None]
        ops.lrt_linear_bwd(dev(x), dev(w_mu), dev(w_rho), dev(b_rho), True, dev(g), var, *outs, eps=dev(eps))
        for name, ours, r64, r32 in zip(["g_x", "g_wmu", "g_wrho", "g_bmu", "g_brho"], outs, g64, g32):
            tol = max(2 * (r32 - r64).abs().max().item(), 3e-6 * r64.abs().max().item())
            # the backward is evaluated at OUR forward's variance (fp32, 1-2 ulp from the fp32 reference's): allow for it
            tol += 1e-6 * r64.abs().max().item()
            assert (ours.cpu().double() - r64).abs().max().item() <= tol, (name, b, i, o, bias)


@unverified("mean_scalars")
def test_r5_sum_scalars_is_the_sequential_fp32_sum(ops):
    """bde_sum_scalars: out = ((s0 + s1) + s2) + ... in fp32 -- bit for bit the loss svgd.py:66,72 accumulates particle by
    particle -- for every count 1..64, scalars that live in separate allocations / inside other tensors, out aliasing an
    input; bad arguments are refused."""
    from beyond_deep_ensembles_amd.ops import BdeKernelError
    rng = np.random.default_rng(7)
    for n in list(range(1, 18)) + [31, 32, 33, 63, 64]:
        vals = (rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)).astype(np.float32)
        block = torch.from_numpy(vals).to(DEV)
        # every other scalar is a view into one block, the others are their own 0-dim tensors
        scalars = [block[i] if i % 2 else torch.tensor(float(vals[i]), dtype=torch.float32, device=DEV) for i in range(n)]
        out = torch.full((), float("nan"), dtype=torch.float32, device=DEV)
        ops.sum_scalars(scalars, out)
        want = np.float32(vals[0])
        for v in vals[1:]:
            want = np.float32(want + v)
        assert out.cpu().numpy() == want, (n, float(out), float(want))
        # bde_mean_scalars: the same sum * fl(1 / divisor) (torch's GPU rounding of svgd.py:105's `/ particle_count`) -- a
        # few particle counts and the scalar count itself
        for div in (1, 3, 5, 8, 13, n):
            mean = torch.full((), float("nan"), dtype=torch.float32, device=DEV)
            ops.mean_scalars(scalars, mean, div)
            expect = want if div == 1 else np.float32(want * (np.float32(1.0) / np.float32(div)))
            assert mean.cpu().numpy() == expect, (n, div)
    a, b = torch.tensor(1.5, device=DEV), torch.tensor([2.25], device=DEV)
    ops.sum_scalars([a, b], a)                                   # out aliases the first input
    assert float(a) == 3.75
    with pytest.raises(BdeKernelError):
        ops.sum_scalars([], a)
    with pytest.raises(BdeKernelError):
        ops.sum_scalars([a] * 65, a)
    with pytest.raises(BdeKernelError):
        ops.sum_scalars([torch.zeros(2, device=DEV)], a)
    with pytest.raises(BdeKernelError):
        ops.sum_scalars([a.double()], a)
    with pytest.raises(BdeKernelError):
        ops.mean_scalars([a, b], a, 0.0)

