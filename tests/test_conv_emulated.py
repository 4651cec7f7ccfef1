"""The fused BBBConv2d kernels' index arithmetic replayed on the CPU (tests/conv_emulator.py) against torch's conv2d and
autograd: forward (bbb_layers.py:146-154), input gradient and weight gradient, with the tilings the library itself
plans.  Runs without a GPU; the kernels themselves are tested by tests/test_ops_gpu.py::test_conv_lrt_*."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import conv_emulator as E

# (N, C, H, W, O, K, stride, padding, bias): small batches of the ResNet-20 layer shapes (so that the planner picks the
# tilings of the real layers: they depend on C, O, H, W, not on N beyond the grid) and ragged geometries
CASES = [
    (2, 3, 32, 32, 16, 3, (1, 1), (1, 1), True), (2, 16, 32, 32, 16, 3, (1, 1), (1, 1), True),
    (2, 16, 32, 32, 32, 3, (2, 2), (1, 1), True), (3, 32, 16, 16, 32, 3, (1, 1), (1, 1), False),
    (2, 32, 16, 16, 64, 3, (2, 2), (1, 1), True), (3, 64, 8, 8, 64, 3, (1, 1), (1, 1), True),
    (2, 16, 32, 32, 32, 1, (2, 2), (0, 0), False),
    (3, 5, 9, 11, 7, 3, (1, 1), (0, 0), True), (2, 7, 13, 6, 33, 3, (2, 1), (1, 2), True), (3, 4, 12, 12, 20, 5, (1, 1), (2, 2), True),
    (1, 3, 30, 30, 40, 7, (2, 2), (3, 3), False), (1, 70, 7, 7, 40, 3, (1, 1), (1, 1), True),
]


def _layer(n, c, h, w, o, k, bias, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, h, w, generator=g)
    x[0, 0, :2] = 0.0
    x[0, 0, 2, : min(w, 3)] = 5e-3
    w_mu, w_rho = torch.randn(o, c, k, k, generator=g) * 0.1, torch.randn(o, c, k, k, generator=g) * 1.5 - 3.0
    w_rho[0, 0] = -8.0
    b_mu, b_rho = (torch.randn(o, generator=g) * 0.1, torch.randn(o, generator=g) - 3.0) if bias else (None, None)
    return x, w_mu, w_rho, b_mu, b_rho


@pytest.mark.parametrize("case", CASES, ids=[f"N{c[0]}C{c[1]}_{c[2]}x{c[3]}_O{c[4]}k{c[5]}s{c[6][0]}{c[6][1]}p{c[7][0]}{c[7][1]}" for c in CASES])
def test_emulated_kernels_equal_torch(case):
    n, c, h, w, o, k, stride, padding, bias = case
    geo = (n, c, h, w, o, k, k, stride[0], stride[1], padding[0], padding[1])
    x, w_mu, w_rho, b_mu, b_rho = _layer(n, c, h, w, o, k, bias, seed=7)
    leaves = [t.double().clone().requires_grad_(True) for t in (x, w_mu, w_rho)]
    xx, wm, wr = leaves
    mean = F.conv2d(xx, wm, None if b_mu is None else b_mu.double(), stride=stride, padding=padding)
    var = F.conv2d((xx ** 2).clamp(min=1e-4), (F.softplus(wr) ** 2).clamp(min=1e-4),
                   None if b_rho is None else F.softplus(b_rho.double()) ** 2, stride=stride, padding=padding)
    g = torch.Generator().manual_seed(3)
    eps, gout = torch.randn(mean.shape, generator=g), torch.randn(mean.shape, generator=g)
    out64 = mean + var.sqrt() * eps.double()
    gx64, gwm64, gwr64 = torch.autograd.grad(out64, leaves, gout.double())

    wb = E.prep(w_mu.numpy(), w_rho.numpy())
    bvar = None if b_rho is None else (F.softplus(b_rho) ** 2).numpy()
    out, var_out = E.conv_kernel(0, geo, x.numpy(), None, wb["wt_mu"], wb["wt_s2"], None if b_mu is None else b_mu.numpy(), bvar,
                                 eps.numpy())
    assert np.isfinite(out).all() and np.isfinite(var_out).all(), "an output element was never written (or read unwritten LDS)"
    np.testing.assert_allclose(var_out, var.detach().numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(out, out64.detach().numpy(), rtol=2e-5, atol=2e-5)

    gvar = (gout * eps / (2.0 * var.detach().sqrt())).float().numpy()
    gx = E.conv_kernel(1, geo, gout.numpy(), gvar, wb["wb_mu"], wb["wb_s2"], x.numpy(), None, None)
    assert np.isfinite(gx).all()
    np.testing.assert_allclose(gx, gx64.numpy(), rtol=3e-5, atol=3e-5 * float(gx64.abs().max()))

    gwm, gwr = E.wgrad_kernel(geo, x.numpy(), gout.numpy(), gvar, w_rho.numpy())
    np.testing.assert_allclose(gwm, gwm64.numpy(), rtol=3e-5, atol=3e-5 * float(gwm64.abs().max()))
    np.testing.assert_allclose(gwr, gwr64.numpy(), rtol=3e-5, atol=3e-5 * float(gwr64.abs().max()))


def test_emulated_kernels_equal_the_reference_layer():
    """The same replay against the fixture written from the REFERENCE's BBBConv2d (tests/golden/conv_lrt.npz): what the
    kernels' index arithmetic produces for the reference's seeded inputs matches the reference's own output and weight /
    input gradients (sums, projections, raw corners) -- for three of the fixture's cases (a ResNet-20 16 -> 16 layer, the
    stride-2 16 -> 32 transition, the ragged 5 x 5 case)."""
    import os
    from oracle.conv_cases import conv_case_inputs, conv_probe_w
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "conv_lrt.npz"), allow_pickle=False)
    for seed, n, c, h, w, o, k, stride, padding, bias in g["cases"].tolist():
        if seed not in (202, 203, 209):
            continue
        x, w_mu, w_rho, b_mu, b_rho, eps, gout, probe_x = conv_case_inputs(seed, n, c, h, w, o, k, stride, padding)
        geo = (n, c, h, w, o, k, k, stride, stride, padding, padding)
        wb = E.prep(w_mu, w_rho)
        bvar = (F.softplus(torch.from_numpy(b_rho)) ** 2).numpy() if bias else None
        out, var = E.conv_kernel(0, geo, x, None, wb["wt_mu"], wb["wt_s2"], b_mu if bias else None, bvar, eps)
        t = f"c{seed}_"
        amax = float(g[t + "out_absmax"])
        np.testing.assert_allclose(out.astype(np.float64).sum((2, 3)), g[t + "out_planes"], atol=3e-5 * amax * np.sqrt(out.shape[2] * out.shape[3]))
        np.testing.assert_allclose(out[0, :, :2, :3], g[t + "out_corner"], atol=3e-5 * amax)
        gvar = (gout * eps / (2.0 * np.sqrt(var))).astype(np.float32)
        gx = E.conv_kernel(1, geo, gout, gvar, wb["wb_mu"], wb["wb_s2"], x, None, None)
        amax = float(g[t + "g_x_absmax"])
        np.testing.assert_allclose(gx.astype(np.float64).sum((2, 3)), g[t + "g_x_planes"], atol=3e-5 * amax * np.sqrt(h * w))
        np.testing.assert_allclose(gx[0, :, :2, :3], g[t + "g_x_corner"], atol=3e-5 * amax)
        gwm, gwr = E.wgrad_kernel(geo, x, gout, gvar, w_rho)
        pw = conv_probe_w(seed, o, c, k).astype(np.float64)
        for name, gw in (("g_wmu", gwm), ("g_wrho", gwr)):
            amax = float(g[t + name + "_absmax"])
            np.testing.assert_allclose(gw.astype(np.float64).sum((1, 2, 3)), g[t + name + "_rowsum"], atol=3e-5 * amax * np.sqrt(c * k * k))
            np.testing.assert_allclose((gw * pw).sum(), float(g[t + name + "_proj"]), atol=3e-5 * amax * np.sqrt(o * c * k * k))
            np.testing.assert_allclose(gw[:2, :2], g[t + name + "_corner"], atol=3e-5 * amax)


def test_planner_supported_means_every_pass_has_a_tiling():
    """bde_conv_lrt_supported (what sends a BBBConv2d layer down the fused path) holds only where ALL three passes have a
    tiling within 64 KB of LDS: forward, input gradient and weight gradient -- over 3000 random geometries and the layer
    shapes of ResNet-20 / -18 / -50 (a wide 1x1 layer on a 56-wide image once had a forward but no weight-gradient plan)."""
    import ctypes
    import random
    from beyond_deep_ensembles_amd import _lib
    lib = _lib.load()
    rng = random.Random(7)
    geos = [(1, 256, 56, 56, 64, 1, 1, 1, 1, 0, 0), (1, 64, 56, 56, 256, 1, 1, 1, 1, 0, 0), (128, 16, 32, 32, 16, 3, 3, 1, 1, 1, 1),
            (1, 3, 224, 224, 64, 7, 7, 2, 2, 3, 3), (2, 512, 7, 7, 512, 3, 3, 1, 1, 1, 1), (1, 1024, 14, 14, 256, 1, 1, 1, 1, 0, 0),
            (1, 2048, 7, 7, 512, 1, 1, 1, 1, 0, 0), (4, 64, 112, 112, 64, 3, 3, 1, 1, 1, 1)]
    for _ in range(3000):
        kh, kw = rng.randint(1, 7), rng.randint(1, 7)
        geos.append((rng.randint(1, 130), rng.choice([1, 3, 16, 33, 64, 256, 700]), rng.randint(kh, 70), rng.randint(kw, 70),
                     rng.choice([1, 10, 16, 64, 100, 512]), kh, kw, rng.randint(1, 3), rng.randint(1, 3), rng.randint(0, kh - 1),
                     rng.randint(0, kw - 1)))
    out = (ctypes.c_int * 16)()
    n_supported = 0
    for geo in geos:
        if not lib.bde_conv_lrt_supported(*geo):
            continue
        n_supported += 1
        for which in (0, 1):
            assert lib.bde_conv_lrt_plan(which, *geo, out) == 0, (which, geo)
            assert 0 < out[14] <= 64 * 1024, (which, geo, out[14])
        assert lib.bde_conv_lrt_bwd_weight_plan(*geo, out) == 0, geo
        assert 0 < out[14] <= 64 * 1024 and 1 <= out[6] <= out[1], (geo, list(out))
        assert lib.bde_conv_lrt_bwd_weight_ws_bytes(*geo) > 0, geo
    assert n_supported > 2500, n_supported                                 # the fused path is the rule, not the exception
    assert all(lib.bde_conv_lrt_supported(*g) for g in geos[:8])


def test_flat_patch_index_split_is_exact():
    """csrc/conv_common.hpp splits a flat patch-plane index e into (row, column) = (e // PWP, e % PWP) with ONE float multiply,
    int((e + 0.5f) * fl(1 / PWP)), in fp32 as the kernel evaluates it.  A plane holds at most 8192 elements (two images of it
    must fit 64 KB of LDS); the split must be exact for every e < 8192 and every row pitch 1 <= PWP <= 8192."""
    e = np.arange(8192, dtype=np.int64)
    ef = e.astype(np.float32) + np.float32(0.5)
    for pwp in range(1, 8193):
        rcp = np.float32(1.0) / np.float32(pwp)
        py = (ef * rcp).astype(np.int32)                  # fp32 product, truncation
        assert np.array_equal(py, e // pwp), pwp


def test_xcd_work_index_is_a_bijection():
    """csrc/conv_common.hpp xcd_work_index: workgroup L (dispatched to XCD L % 8) -> a work index such that one XCD's workgroups
    own CONSECUTIVE work indices; it must be a bijection on [0, total) for every total."""
    for total in list(range(1, 70)) + [255, 256, 257, 511, 512, 770, 1000]:
        q, r = divmod(total, 8)
        work = [(L % 8) * q + min(L % 8, r) + L // 8 for L in range(total)]
        assert sorted(work) == list(range(total)), total
        for x in range(min(8, total)):                               # the workgroups of XCD x: one contiguous range
            mine = sorted(w for L, w in enumerate(work) if L % 8 == x)
            assert mine == list(range(mine[0], mine[0] + len(mine))), (total, x)
