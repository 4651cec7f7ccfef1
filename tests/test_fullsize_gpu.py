"""Parity at BASELINE.json's full sizes (ResNet-50-sized weights, M = 8, K = 20, S = 30) on a real
MI355X.  At this size the CPU oracle is used once (a few seconds); the rest are size-independent
properties of the path: determinism, translation invariance and permutation equivariance of the SVGD
direction, linearity in the gradients, fp64 re-evaluation of the closed forms with torch on the GPU,
SWAG moments == running averages of the iterates, ring <-> iterate mapping, sample linearity in the
noise, batched == unbatched sampling."""
import math

import numpy as np
import pytest
import torch

from oracle import bde_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 23_880_950
M, K, S = 8, 20, 30
LD = (D + 16 + 63) // 64 * 64


@pytest.fixture(scope="module")
def ops():
    from beyond_deep_ensembles_amd.ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def svgd_inputs():
    g = torch.Generator(device=DEV).manual_seed(1234)
    P = torch.zeros(M, LD, device=DEV)
    theta0 = torch.randn(D, device=DEV, generator=g) * 0.05
    P[:, :D] = theta0                                        # shared backbone (iwildcam/models.py:118-119)
    head = 372_918
    P[:, D - head:D] += (torch.rand(M, head, device=DEV, generator=g) * 2 - 1) / math.sqrt(2048)
    G = torch.zeros(M, LD, device=DEV)
    G[:, :D] = torch.randn(M, D, device=DEV, generator=g) * 0.01
    return P, G


def phi64_gpu(P, G, l2, scale, n):
    """fp64 evaluation of svgd.py:86-89 with torch on the GPU (independent of libbde_hip)."""
    p, g = P[:, :D].double(), G[:, :D].double()
    d2 = torch.cdist(p, p, p=2) ** 2
    h = torch.sqrt(0.5 * torch.quantile(d2, 0.5) / math.log(M + 1)) + 1e-8
    k = torch.exp(-d2 / (2 * h ** 2))
    gk = (k.sum(1, keepdim=True) * p - k @ p) / h ** 2
    return k @ (-(g + l2 / 2 * p)) + scale * gk / n, k


def run(ops, P, G, l2=0.0, scale=1.0, n=129809.0):
    out = torch.empty_like(G)
    ws, ks = ops.svgd_ws(M, DEV), ops.svgd_kstat(M, DEV)
    ops.svgd_step(P, G, out, D, l2, scale, n, -1.0, ws, ks)
    torch.cuda.synchronize()
    return out, ks


def test_svgd_fullsize_vs_oracle_and_fp64(ops, svgd_inputs):
    P, G = svgd_inputs
    out, ks = run(ops, P, G, l2=1e-5)
    phi64, k64 = phi64_gpu(P, G, 1e-5, 1.0, 129809.0)
    ref32 = O.svgd_phi(P[:, :D].cpu(), G[:, :D].cpu(), 1e-5, 1.0, 129809.0).to(DEV)     # the CPU oracle, once
    err = (-out[:, :D].double() - phi64).abs().max().item()
    err_ref = (ref32.double() - phi64).abs().max().item()
    mag = phi64.abs().max().item()
    assert err <= max(2 * err_ref, 3e-6 * mag), (err, err_ref, mag)
    kmat = ks[:M * M].view(M, M).double()
    assert (kmat - k64).abs().max().item() <= 5e-6
    out2, _ = run(ops, P, G, l2=1e-5)
    assert torch.equal(out[:, :D], out2[:, :D])                               # run-to-run deterministic


def test_svgd_fullsize_properties(ops, svgd_inputs):
    P, G = svgd_inputs
    base, ks = run(ops, P, G)
    mag = base[:, :D].abs().max().item()
    # translation invariance (l2_reg = 0): adding one vector to every particle changes neither K nor phi
    c = torch.randn(D, device=DEV) * 0.01
    Pt = P.clone()
    Pt[:, :D] += c
    shifted, ks_t = run(ops, Pt, G)
    assert (ks_t[:M * M] - ks[:M * M]).abs().max().item() <= 2e-5
    assert (shifted[:, :D] - base[:, :D]).abs().max().item() <= 2e-5 * mag
    # permutation equivariance: permuting the particles permutes the rows of phi
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device=DEV)
    permuted, _ = run(ops, P[perm].contiguous(), G[perm].contiguous())
    assert (permuted[:, :D] - base[perm][:, :D]).abs().max().item() <= 2e-6 * mag
    # linearity in the gradients: phi(P, G1 + G2) + phi(P, 0) == phi(P, G1) + phi(P, G2)
    G2 = torch.zeros_like(G)
    G2[:, :D] = torch.randn(M, D, device=DEV) * 0.01
    both, _ = run(ops, P, G + G2)
    zero, _ = run(ops, P, torch.zeros_like(G))
    only2, _ = run(ops, P, G2)
    assert ((both + zero) - (base + only2))[:, :D].abs().max().item() <= 4e-6 * mag


def test_svgd_fused_fullsize(ops, svgd_inputs):
    """One pass (phi + M shared-state SGD steps + next Gram) == combine followed by apply, at full size."""
    P, G = svgd_inputs
    ws, ks, wsn = ops.svgd_ws(M, DEV), ops.svgd_kstat(M, DEV), ops.svgd_ws(M, DEV)
    ops.svgd_gram(P, D, ws)
    ops.svgd_kstats(ws, M, 0.0, 1.0, 129809.0, -1.0, ks)
    Pa, Pb = P.clone(), P.clone()
    tmp = torch.empty_like(G)
    ba, bb = torch.zeros(LD, device=DEV), torch.zeros(LD, device=DEV)
    ops.svgd_combine(Pa, G, tmp, D, ks)
    ops.svgd_apply_sgd(Pa, tmp, ba, D, 0.05, 0.9, 0.0, 3e-4, True, True)
    ops.svgd_fused_sgd(Pb, G, bb, D, ks, 0.05, 0.9, 0.0, 3e-4, True, True, ws_next=wsn)
    assert (Pa[:, :D] - Pb[:, :D]).abs().max().item() <= 1e-7
    ks2, ks3 = ops.svgd_kstat(M, DEV), ops.svgd_kstat(M, DEV)
    ops.svgd_gram(Pb, D, ws)
    ops.svgd_kstats(ws, M, 0.0, 1.0, 129809.0, -1.0, ks2)
    ops.svgd_kstats(wsn, M, 0.0, 1.0, 129809.0, -1.0, ks3)        # from the fused kernel's Gram partials
    assert (ks2[:M * M] - ks3[:M * M]).abs().max().item() <= 2e-5


def test_swag_fullsize(ops):
    g = torch.Generator(device=DEV).manual_seed(7)
    theta = torch.zeros(LD, device=DEV)
    theta[:D] = torch.randn(D, device=DEV, generator=g) * 0.05
    mean, sq = theta.clone(), theta * theta
    ring = torch.zeros(K, LD, device=DEV)
    head = 0
    run_sum, run_sq = theta[:D].double().clone(), (theta[:D].double()) ** 2
    last = {}
    # the CPU oracle replays the same 25 updates on two slices (first 2^20 and last 1003 parameters)
    slices = (slice(0, 1 << 20), slice(D - 1003, D))
    oracle_states = [O.swag_init(theta[sl].cpu(), K) for sl in slices]
    for n in range(1, 26):                                                   # 25 updates: every ring row written
        theta[:D] += torch.randn(D, device=DEV, generator=g) * 1e-3
        ops.swag_update(theta, mean, sq, ring[head], n, D)
        for st, sl in zip(oracle_states, slices):
            st.updates = n
            O.swag_moment_update(st, theta[sl].cpu())
        run_sum += theta[:D].double()
        run_sq += theta[:D].double() ** 2
        last[head] = (n, theta[:D].clone(), mean[:D].clone())
        head = (head + 1) % K
    # moments == running averages of the iterates (initial weights are sample #1, swag.py:32)
    assert (mean[:D].double() - run_sum / 26).abs().max().item() <= 2e-7
    assert (sq[:D].double() - run_sq / 26).abs().max().item() <= 2e-8
    # ring <-> iterate mapping, bit-exact: row r holds theta_t - mean_t of the update that wrote it
    for r, (n, th, mn) in last.items():
        assert torch.equal(ring[r, :D], th - mn), (r, n)
    # logical column K-1 (newest) is physical row head-1
    assert last[(head - 1) % K][0] == 25
    # ... and against the oracle (the reference's fp32 op order, swag.py:98-104): moments and all K columns BIT-EXACT
    order0 = [(head + c) % K for c in range(K)]
    for st, sl in zip(oracle_states, slices):
        assert torch.equal(mean[sl].cpu(), st.mean) and torch.equal(sq[sl].cpu(), st.sq_weights)
        assert torch.equal(ring[order0][:, sl].t().cpu(), st.deviations)
    # sampling: eps = 0 returns the mean; linear in the noise; batched == unbatched; fp64 closed form
    zeros_w, zeros_d = torch.zeros(K, device=DEV), torch.zeros(LD, device=DEV)
    out0, out1, out2, out12 = (torch.empty(LD, device=DEV) for _ in range(4))
    ops.swag_sample(mean, sq, ring, head, out0, D, eps_w=zeros_w, eps_d=zeros_d)
    assert torch.equal(out0[:D], mean[:D])
    e1w, e2w = torch.randn(K, device=DEV, generator=g), torch.randn(K, device=DEV, generator=g)
    e1d, e2d = torch.randn(LD, device=DEV, generator=g), torch.randn(LD, device=DEV, generator=g)
    ops.swag_sample(mean, sq, ring, head, out1, D, eps_w=e1w, eps_d=e1d)
    ops.swag_sample(mean, sq, ring, head, out2, D, eps_w=e2w, eps_d=e2d)
    ops.swag_sample(mean, sq, ring, head, out12, D, eps_w=e1w + e2w, eps_d=e1d + e2d)
    assert ((out12 + out0) - (out1 + out2))[:D].abs().max().item() <= 3e-6
    order = [(head + c) % K for c in range(K)]
    # closed form: the low-rank term in fp64, the diagonal term with the reference's fp32 op order
    # (sq - mean**2 cancels in fp32 in the reference too, swag.py:112 -- that rounding is part of the spec)
    lowrank = (ring[order, :D].double().t() / math.sqrt(2 * (K - 1))) @ e1w.double()
    sd32 = (0.5 * (torch.relu(sq[:D] - mean[:D] ** 2) + 1e-6)).sqrt()
    want = mean[:D].double() + lowrank + sd32.double() * e1d[:D].double()
    assert (out1[:D].double() - want).abs().max().item() <= 2e-7
    # the oracle's sample (swag.py:57,112-114: LowRankMultivariateNormal arithmetic) on the slices, same noise
    for st, sl in zip(oracle_states, slices):
        ref = O.swag_sample(st.mean, st.sq_weights, st.deviations, e1w.cpu(), e1d[sl].cpu())
        assert (out1[sl].cpu() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    # S = 30 in one pass (MFMA) == 30 single samples, Philox noise (stream = sample index)
    outb = torch.empty(S, LD, device=DEV)
    ops.swag_sample_batched(mean, sq, ring, head, outb, D, seed=5, stream_id0=100)
    for s in (0, 13, 29):
        ops.swag_sample(mean, sq, ring, head, out1, D, seed=5, stream_id=100 + s)
        assert (outb[s, :D] - out1[:D]).abs().max().item() <= 1e-6
    z = (outb[:, :D] - mean[:D]).flatten()[::97]
    assert torch.isfinite(z).all()


def test_bbb_ivon_fullsize(ops):
    g = torch.Generator(device=DEV).manual_seed(11)
    mean = torch.zeros(LD, device=DEV)
    mean[:D] = torch.randn(D, device=DEV, generator=g) * 0.1                  # blundell_init (util.py:161-163)
    rho = torch.full((LD,), -3.0, device=DEV)
    rho[:D] += torch.randn(D, device=DEV, generator=g) * 0.5
    ws = ops.reduce_ws(DEV)
    kl = torch.zeros(1, device=DEV)
    gm, gr = torch.empty(LD, device=DEV), torch.empty(LD, device=DEV)
    ops.gauss_kl(mean, rho, 0.0, 1.0, D, ws, kl_out=kl, gmean=gm, grho=gr, grad_scale=1.0)
    m64, s64 = mean[:D].double(), torch.nn.functional.softplus(rho[:D].double())
    want = (0.5 * (2 * torch.log(1.0 / s64) - 1 + s64 ** 2 + m64 ** 2)).sum().item()      # bbb.py:20, fp64
    assert abs(kl.item() - want) <= 2e-6 * abs(want)
    assert (gm[:D].double() - m64).abs().max().item() <= 1e-6
    want_gr = (-1 / s64 + s64) * torch.sigmoid(rho[:D].double())
    assert ((gr[:D].double() - want_gr).abs() / (1 / s64 + s64)).max().item() <= 3e-6
    # the oracle (autograd of bbb.py:20 / util.py:171 in fp32) on the first 2^20 and the last 1003 parameters
    for sl in (slice(0, 1 << 20), slice(D - 1003, D)):
        ogm, ogr = O.gauss_kl_grads(mean[sl].cpu(), rho[sl].cpu(), 0.0, 1.0)
        assert (gm[sl].cpu() - ogm).abs().max().item() <= 1e-6
        assert ((gr[sl].cpu() - ogr).abs() / (1 / s64[sl].cpu().float() + s64[sl].cpu().float())).max().item() <= 3e-6
    # draw: Philox forward == supplied-noise forward; backward regenerates the same noise
    w1, w2, eps = (torch.empty(LD, device=DEV) for _ in range(3))
    ops.gauss_draw_fwd(mean, rho, w1, D, seed=3, stream_id=9, eps_out=eps)
    ops.gauss_draw_fwd(mean, rho, w2, D, eps=eps)
    assert torch.equal(w1[:D], w2[:D])
    assert (w1[:D].double() - (m64 + s64 * eps[:D].double())).abs().max().item() <= 1e-6
    for sl in (slice(0, 1 << 20), slice(D - 1003, D)):
        ref = O.gauss_sample(mean[sl].cpu(), rho[sl].cpu(), eps[sl].cpu())
        assert (w1[sl].cpu() - ref).abs().max().item() <= 1e-6
    assert abs(eps[:D].mean().item()) < 1e-3 and abs(eps[:D].var().item() - 1) < 1e-3
    # iVON update vs the fp64 closed form (ivorn.py:79-89)
    prec = torch.full((LD,), 100.0 / 129809.0, device=DEV)
    prec[:D] *= (1 + torch.rand(D, device=DEV, generator=g))
    mom = torch.zeros(LD, device=DEV)
    mom[:D] = torch.randn(D, device=DEV, generator=g) * 1e-3
    dsum, acc = torch.zeros(LD, device=DEV), torch.zeros(LD, device=DEV)
    dsum[:D] = torch.randn(D, device=DEV, generator=g) * 0.01
    acc[:D] = torch.randn(D, device=DEV, generator=g) * 0.02
    m2, mo2, p2 = mean.clone(), mom.clone(), prec.clone()
    ops.ivon_update(m2, mo2, p2, dsum, acc, D, lam=100.0 / 129809.0, n_eff=129809.0, mc=2, beta1=0.9, beta2=0.999,
                    t=3, lr=1e-3, damping=1e-3)
    # bit-exact against the CPU oracle (fp32, the reference's op order) on the first 2^20 and the last 1003 elements
    for sl in (slice(0, 1 << 20), slice(D - 1003, D)):
        want = O.ivon_update(mean[sl].cpu(), mom[sl].cpu(), prec[sl].cpu(), dsum[sl].cpu(), acc[sl].cpu(), step_t=3,
                             lr=1e-3, prior_prec=100.0, dataset_size=129809.0, damping=1e-3, mc_samples=2)
        for got, w in zip((m2, mo2, p2), want):
            assert torch.equal(got[sl].cpu(), w)


# ---- the other BASELINE.json configs as parity cases -----------------------------------------------
@pytest.mark.parametrize("name,d", [("cifar_resnet20", 273_610), ("camelyon_densenet121", 6_955_906)])
def test_baseline_config_sizes_svgd_and_swag(ops, name, d):
    """configs[1]/[2] (CIFAR-10 ResNet-20: SVGD 8 particles; SWAG K=20, 30 samples) and configs[4]
    (Camelyon17 DenseNet-121 MultiSWAG: 5 modes x 30 samples) at their real parameter counts, HIP vs the CPU oracle."""
    ld = (d + 16 + 63) // 64 * 64
    g = torch.Generator().manual_seed(3)
    P = torch.randn(M, d, generator=g) * 0.05
    G = torch.randn(M, d, generator=g) * 0.01
    Pb, Gb = torch.zeros(M, ld, device=DEV), torch.zeros(M, ld, device=DEV)
    Pb[:, :d], Gb[:, :d] = P.to(DEV), G.to(DEV)
    ws, ks = ops.svgd_ws(M, DEV), ops.svgd_kstat(M, DEV)
    ops.svgd_step(Pb, Gb, Gb, d, 3e-4, 1.0, 50000.0, -1.0, ws, ks)            # cifar.yaml-like: l2 3e-4, N 50,000
    phi64 = O.svgd_phi(P.double(), G.double(), 3e-4, 1.0, 50000.0)
    ref32 = O.svgd_phi(P, G, 3e-4, 1.0, 50000.0)
    err = (-Gb[:, :d].cpu().double() - phi64).abs().max().item()
    err_ref = (ref32.double() - phi64).abs().max().item()
    assert err <= max(2 * err_ref, 3e-6 * phi64.abs().max().item()), (name, err, err_ref)
    # SWAG: 25 updates, then samples of every "mode" with supplied noise vs the oracle; batched == unbatched
    modes = 5 if "densenet" in name else 1
    for mode in range(modes):
        theta = torch.randn(d, generator=g) * 0.05
        st = O.swag_init(theta, K)
        mean, sq = torch.zeros(ld, device=DEV), torch.zeros(ld, device=DEV)
        mean[:d], sq[:d] = st.mean.to(DEV), st.sq_weights.to(DEV)
        ring = torch.zeros(K, ld, device=DEV)
        th = torch.zeros(ld, device=DEV)
        head = 0
        for n in range(1, 26 if mode == 0 else 4):
            theta = theta + torch.randn(d, generator=g) * 1e-3
            st.updates = n
            O.swag_moment_update(st, theta)
            th[:d] = theta.to(DEV)
            ops.swag_update(th, mean, sq, ring[head], n, d)
            head = (head + 1) % K
        assert torch.equal(mean[:d].cpu(), st.mean) and torch.equal(sq[:d].cpu(), st.sq_weights)
        ew, ed = torch.randn(S, K, generator=g), torch.randn(S, d, generator=g)
        edb = torch.zeros(S, ld, device=DEV)
        edb[:, :d] = ed.to(DEV)
        outb = torch.zeros(S, ld, device=DEV)
        ops.swag_sample_batched(mean, sq, ring, head, outb, d, eps_w=ew.to(DEV), eps_d=edb)
        out1 = torch.zeros(ld, device=DEV)
        for s in (0, S - 1):
            want = O.swag_sample(st.mean, st.sq_weights, st.deviations, ew[s], ed[s])
            ops.swag_sample(mean, sq, ring, head, out1, d, eps_w=ew[s].to(DEV), eps_d=edb[s])
            assert torch.allclose(out1[:d].cpu(), want, rtol=2e-5, atol=2e-6), (name, mode, s)
            assert torch.allclose(outb[s, :d].cpu(), want, rtol=2e-5, atol=2e-6), (name, mode, s)
