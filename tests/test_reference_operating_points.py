"""The optimizers at the keyword values the reference's own experiment files use ("kwarg values actually
exercised"), next to the IMPORTED reference on the same seeds, data and noise (build container only: skipped where
/root/reference is absent, e.g. on the GPU box).

Two backends behind our shells: "oracle" (the CPU checker) and "emu" -- the product's HipOps over the kernel SOURCES compiled
for the CPU execution model of tests/hip_emu, so the comparison is reference -> C ABI -> kernels without the oracle in
between.  What a default-constructed object launches (the streaming SVGD kernels, the SWAG / iVON kernels) is what runs.

Operating points (reference file:line):
  SVGD   particle_count 5, kernel_grad_scale 1.0, l2_reg 0.0 / dataset_size 129809 over Adam(lr 3e-5)  iwildcam.yaml:213-221
         l2_reg 0.01 / dataset_size 269038 over Adam(lr 1e-3)                                          uci.yaml:234-242
         SGD(lr .05, momentum .9, nesterov, wd 3e-4), dataset_size 50000                               cifar.yaml:218-223
  SWAG   deviation_samples 30 (iwildcam.yaml:97, cifar.yaml:72) and 10 (civil.yaml), start_epoch > 0
  iVON   lr 1e-4, prior_prec 50, damping 1e-3, augmentation 10, mc_samples 2, dataset_size 50000       cifar.yaml:163-168
         lr 3e-5, prior_prec 100, damping 1e-3, augmentation 1, mc_samples 2, dataset_size 129809      iwildcam.yaml:199-204
Tolerances are the ones of tests/test_against_reference_import.py: integer schedule and SWAG moments bit-exact, SVGD rtol
1e-5 (fp32 reduction order of the Gram / kernel sums differs from ATen's).  iVON: the checker (ATen on this CPU) is
bit-identical to the reference; the KERNEL is bit-identical to the IEEE-754 sequence of ivorn.py:108 (pinned against numpy
below) and therefore up to one ulp of sqrt away from this container's reference run: torch's CPU `sqrt` is MKL's vmsSqrt in
VML_HA mode (< 1 ulp, not correctly rounded; e.g. sqrt(fl32 0x1.58e2d4p+9) = 0x1.a43758p+4, correctly rounded
0x1.a4375ap+4), which shows up in about one noise element in 400 and from there at the 1e-7 level in the trajectory."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "algos")), reason="reference checkout absent")


@pytest.fixture(scope="module")
def ref():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    try:
        import src.algos.svgd as svgd
        import src.algos.swag as swag
        import src.algos.ivorn as ivon
        import src.algos.util as util
    finally:
        sys.path.remove(REF)
    return {"svgd": svgd, "swag": swag, "ivon": ivon, "util": util}


@pytest.fixture(params=["oracle", "emu"])
def backend(request):
    if request.param == "oracle":
        from tests.oracle_ops import OracleOps
        yield OracleOps()
        return
    from tests.hip_emu import build, emu_ops
    if not build.available():
        pytest.skip("no host clang / HIP headers to build the CPU model with")
    with emu_ops.emulated(emu_ops.ALL) as ops:
        yield ops


def mlp(seed, hidden=9):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(6, hidden), nn.Tanh(), nn.Linear(hidden, 2))


def flat(ps):
    return torch.cat([p.detach().reshape(-1) for p in ps])


def data(seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(24, 6, generator=g), torch.randn(24, 2, generator=g)


SVGD_POINTS = {
    "iwildcam": dict(l2_reg=0.0, dataset_size=129809, base=lambda ps: torch.optim.Adam(ps, lr=3e-5, weight_decay=0)),
    "uci": dict(l2_reg=0.01, dataset_size=269038, base=lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=0)),
    "cifar": dict(l2_reg=3e-4, dataset_size=50000,
                  base=lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4)),
}


@pytest.mark.parametrize("path", ["default", pytest.param("small_model", marks=pytest.mark.device_unverified(
    "svgd_small", "small_step_host", "mean_scalars", "fast_loop"))])
@pytest.mark.parametrize("hidden", [9, 601])        # D = 83 / 5,411 in four tensors (ragged: 16-byte padding behind three of them)
@pytest.mark.parametrize("point", sorted(SVGD_POINTS))
def test_svgd_at_the_reference_yaml_values(ref, backend, point, hidden, path):
    """svgd.py:44-105 with particle_count 5 / kernel_grad_scale 1.0 and each experiment's l2_reg, dataset_size and base
    optimizer: four steps, every particle row after the last one and every returned loss.  The wide case scales its loss by
    0.1: with lr 0.05 and the raw MSE of 601 hidden units the steps are as large as the weights (0.5 - 1.0 per step) and the
    training loop itself amplifies a first-step difference of 1.3e-7 to 6e-6 by step four -- in the reference against a
    re-run of itself in another summation order just the same; scaled, every backend stays within one ulp of the reference.
    path "small_model": the opt-in step of DESIGN section 0 (the two-launch small-model kernel, the native host paths, the
    one-launch loss mean) instead of what a default-constructed optimizer runs -- never run on a device, on the CPU model here."""
    import beyond_deep_ensembles_amd as bde
    torch.set_num_threads(1)
    cfg = SVGD_POINTS[point]
    x, y = data(11)
    ls = 1.0 if hidden == 9 else 0.1
    results = []
    for which in ("ref", "ours"):
        model = mlp(3, hidden)
        torch.manual_seed(103)             # the reset closure consumes the same RNG stream in both runs
        kw = dict(particle_count=5, dataset_size=cfg["dataset_size"], l2_reg=cfg["l2_reg"], kernel_grad_scale=1.0)
        if which == "ref":
            opt = ref["svgd"].SVGDOptimizer(model.parameters(), lambda: ref["util"].reset_model_params(model),
                                            cfg["base"](model.parameters()), **kw)
        else:
            opt = bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model),
                                    cfg["base"](model.parameters()), _ops=backend,
                                    **(dict(single_launch="two", host_fast_paths=True) if path == "small_model" else {}), **kw)
        losses = []
        for t in range(4):
            xb, yb = x[(t % 3) * 8:(t % 3 + 1) * 8], y[(t % 3) * 8:(t % 3 + 1) * 8]
            losses.append(float(opt.step(lambda: ls * F.mse_loss(model(xb), yb), lambda l: l.backward())))
        params = list(model.parameters())
        parts = torch.stack([flat([opt.state[p][f"particle_{i}"] for p in params]) for i in range(5)])
        results.append((parts, losses))
    np.testing.assert_allclose(results[1][0].numpy(), results[0][0].numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(results[1][1], results[0][1], rtol=1e-6)


@pytest.mark.parametrize("k,start_epoch,interval", [(30, 1, 1), (10, 2, 2), (30, 0, 3.7)])
def test_swag_at_the_reference_yaml_values(ref, backend, k, start_epoch, interval):
    """swag.py:15-114 with deviation_samples 30 / 10: fewer updates than columns (the ring is partly zero, Q9: the factor stays
    sqrt(2 (K - 1))) and, for K = 10, more updates than columns (the ring wraps).  Schedule and moments bit-exact; the sample
    with the reference's random stream."""
    import beyond_deep_ensembles_amd as bde
    torch.set_num_threads(1)
    x, y = data(21)
    outs = []
    for which in ("ref", "ours"):
        model = mlp(5, hidden=5)
        base = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4)
        kw = dict(update_interval=interval, start_epoch=start_epoch, deviation_samples=k)
        opt = ref["swag"].SwagOptimizer(model.parameters(), base, **kw) if which == "ref" else \
            bde.SwagOptimizer(model.parameters(), base, _ops=backend, **kw)
        for epoch in range(5):
            for t in range(6):
                xb, yb = x[(t % 3) * 8:(t % 3 + 1) * 8], y[(t % 3) * 8:(t % 3 + 1) * 8]
                opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())
            opt.complete_epoch()
        torch.manual_seed(7)
        opt.sample_parameters()
        sample = flat(list(model.parameters()))
        opt.step(lambda: F.mse_loss(model(x[:8]), y[:8]), lambda l: l.backward())      # restores, then steps
        after = flat(list(model.parameters()))
        if which == "ref":
            stats = (opt.state["__mean"], opt.state["__sq_weights"], opt.state["__deviations"])
        else:
            stats = (opt.mean_vector(), opt.sq_vector(), opt.deviations_dk())
        counters = (opt.state["__epoch"], opt.state["__steps_since_swag_start"], opt.state["__updates"])
        outs.append(([s.cpu() for s in stats], counters, sample, after))
    assert outs[0][1] == outs[1][1]
    assert outs[0][1][2] > 0
    for a, b in zip(outs[0][0], outs[1][0]):
        np.testing.assert_array_equal(a.numpy(), b.numpy())
    # the sample: mean + W eps_W + sqrt(diag) eps_D on the reference's noise.  The checker repeats ATen's order of operations;
    # the kernel sums the K low-rank terms in its own order (tests/test_ops_gpu.py::test_swag_sample_golden_and_oracle: 1e-5)
    np.testing.assert_allclose(outs[1][2].numpy(), outs[0][2].numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(outs[0][3].numpy(), outs[1][3].numpy())


@pytest.mark.parametrize("point,kw", [
    ("cifar", dict(lr=1e-4, prior_prec=50, damping=1e-3, augmentation=10, mc_samples=2, dataset_size=50000)),
    ("iwildcam", dict(lr=3e-5, prior_prec=100, damping=1e-3, augmentation=1, mc_samples=2, dataset_size=129809)),
    ("uci", dict(lr=1e-3, prior_prec=1, damping=1e-3, augmentation=1, mc_samples=5, dataset_size=455))])
def test_ivon_at_the_reference_yaml_values(ref, backend, point, kw):
    """ivorn.py:15-127 at the experiments' values: four steps, mean / precision / live parameters bit-identical."""
    import beyond_deep_ensembles_amd as bde
    torch.set_num_threads(1)
    x, y = data(31)
    outs = []
    for which in ("ref", "ours"):
        model = mlp(7, hidden=5)
        opt = ref["ivon"].iVONOptimizer(model.parameters(), **kw) if which == "ref" else \
            bde.iVONOptimizer(model.parameters(), _ops=backend, **kw)
        torch.manual_seed(55)            # rng="torch": one normal_like per tensor, in parameter order, as the reference
        losses = []
        for t in range(4):
            xb, yb = x[(t % 3) * 8:(t % 3 + 1) * 8], y[(t % 3) * 8:(t % 3 + 1) * 8]
            losses.append(float(opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())))
        params = list(model.parameters())
        outs.append((flat([opt.state[p]["mean"] for p in params]), flat([opt.state[p]["precision"] for p in params]),
                     flat(params), losses))
    if type(backend).__name__ == "OracleOps":
        for a, b in zip(outs[0][:3], outs[1][:3]):
            np.testing.assert_array_equal(a.numpy(), b.numpy())
        assert outs[0][3] == outs[1][3]
    else:           # the kernels: IEEE sqrt against MKL's VML_HA sqrt in the reference's sample (module docstring)
        for a, b in zip(outs[0][:3], outs[1][:3]):
            np.testing.assert_allclose(b.numpy(), a.numpy(), rtol=2e-6, atol=2e-7)    # atol: mean + delta cancels in some live parameters
        np.testing.assert_allclose(outs[1][3], outs[0][3], rtol=1e-6)


@pytest.mark.parametrize("n_eff", [129809.0, 500000.0])
def test_ivon_sample_kernel_is_the_ieee_sequence_of_the_reference_line(n_eff):
    """ivorn.py:108 `1 / (N * precision.clamp(min=1e-4)).sqrt() * noise`, then :111 `mean + delta`, evaluated in numpy fp32
    (multiply, correctly rounded sqrt, divide, multiply, add -- one rounding each): the kernel source on the CPU model equals
    it bit for bit on 100,003 elements (ragged tail included), first and accumulating call.  This is the pin that does not
    depend on which sqrt a torch build links."""
    from tests.hip_emu import build, emu_ops
    if not build.available():
        pytest.skip("no host clang / HIP headers to build the CPU model with")
    rng = np.random.default_rng(int(n_eff))
    n = 100003
    mean = rng.standard_normal(n).astype(np.float32) * np.float32(0.3)
    prec = np.exp(rng.uniform(np.log(2e-5), np.log(0.5), n)).astype(np.float32)       # both sides of the 1e-4 clamp
    eps = rng.standard_normal(n).astype(np.float32)
    f = np.float32
    delta = (f(1) / np.sqrt(f(n_eff) * np.maximum(prec, f(1e-4)))) * eps
    with emu_ops.emulated(["ivon.hip"]) as ops:
        param, dsum = torch.empty(n), torch.empty(n)
        ops.ivon_sample(torch.from_numpy(mean), torch.from_numpy(prec), param, dsum, n, n_eff, True, eps=torch.from_numpy(eps))
        np.testing.assert_array_equal(param.numpy(), mean + delta)
        np.testing.assert_array_equal(dsum.numpy(), delta)
        ops.ivon_sample(torch.from_numpy(mean), torch.from_numpy(prec), param, dsum, n, n_eff, False, eps=torch.from_numpy(eps))
        np.testing.assert_array_equal(dsum.numpy(), delta + delta)


BBB_POINTS = {
    # prior_std 1.0, mc_samples 2, kl_rescaling 0.2, dataset_size 50000, SGD(momentum .9, nesterov, wd 0)   cifar.yaml:125-135
    "cifar": dict(prior=1.0, mc=2, klr=0.2, n=50000,
                  base=lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=0.0)),
    # prior_std 1.0, mc_samples 2, kl_rescaling 1.0, dataset_size 129809, Adam(lr 3e-5... here 1e-3), wd 0   iwildcam.yaml:136-143
    "iwildcam": dict(prior=1.0, mc=2, klr=1.0, n=129809, base=lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=0.0)),
    # mc_samples 5, the script's dataset_size (the housing split: 455 rows), Adam                              uci.yaml:160-168
    "uci": dict(prior=1.0, mc=5, klr=1.0, n=455, base=lambda ps: torch.optim.Adam(ps, lr=1e-3)),
    # prior_std 0.1, kl_rescaling 1.0                                                                          cifar.yaml:236-246
    "cifar_narrow_prior": dict(prior=0.1, mc=2, klr=1.0, n=50000,
                               base=lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4)),
}


@pytest.mark.parametrize("point", sorted(BBB_POINTS))
def test_bbb_at_the_reference_yaml_values(ref, backend, point):
    """bbb.py:47-89 over weight-sampling layers built on each side's GaussianParameter (util.py:151-186), the same noise tape
    on both sides, blundell_init (util.py:161-163: mean ~ N(0, 0.1^2), rho = -3): four steps, every loss and every mean / rho."""
    import beyond_deep_ensembles_amd as bde
    sys.path.insert(0, REF)
    try:
        import src.algos.bbb as rbbb
    finally:
        sys.path.remove(REF)
    torch.set_num_threads(1)
    cfg = BBB_POINTS[point]
    mc = cfg["mc"]
    g = torch.Generator().manual_seed(77)
    tape = [torch.randn(s, generator=g) for _ in range(4 * mc) for s in ((7, 6), (7,), (2, 7), (2,))]
    x, y = data(41)

    def build(side):
        noise = [t.clone() for t in tape]
        GP = ref["util"].GaussianParameter if side == "ref" else (lambda size: bde.GaussianParameter(size, _ops=backend))

        class Lin(nn.Module):
            def __init__(self, i, o):
                super().__init__()
                self.weight, self.bias = GP((o, i)), GP((o,))

            def forward(self, inp):
                return F.linear(inp, self.weight.sample(), self.bias.sample())
        model = nn.Sequential(Lin(6, 7), nn.Tanh(), Lin(7, 2))
        with torch.no_grad():
            for p in model.parameters():
                if getattr(p, "_is_gaussian_rho", False):
                    p.fill_(-3.0)
                else:
                    p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(5 + p.numel())) * 0.1)
        params = list(model.parameters())
        if side == "ref":
            ref["util"].normal_like = lambda t: noise.pop(0)
            opt = rbbb.BBBOptimizer(params, cfg["base"](params), rbbb.GaussianPrior(0.0, cfg["prior"]), dataset_size=cfg["n"],
                                    mc_samples=mc, kl_rescaling=cfg["klr"])
        else:
            for mod in model.modules():
                if isinstance(mod, bde.GaussianParameter):
                    mod.noise_source = lambda rho: noise.pop(0)
            opt = bde.BBBOptimizer(params, cfg["base"](params), bde.GaussianPrior(0.0, cfg["prior"]), dataset_size=cfg["n"],
                                   mc_samples=mc, kl_rescaling=cfg["klr"], _ops=backend)
        return model, params, opt

    old = ref["util"].normal_like
    try:
        m_r, p_r, o_r = build("ref")
        losses_r, traj_r = [], []
        for t in range(4):
            xb, yb = x[(t % 3) * 8:(t % 3 + 1) * 8], y[(t % 3) * 8:(t % 3 + 1) * 8]
            losses_r.append(float(o_r.step(lambda: F.mse_loss(m_r(xb), yb), lambda l: l.backward()).detach()))
            traj_r.append(flat(p_r))
    finally:
        ref["util"].normal_like = old
    m_o, p_o, o_o = build("ours")
    for t in range(4):
        xb, yb = x[(t % 3) * 8:(t % 3 + 1) * 8], y[(t % 3) * 8:(t % 3 + 1) * 8]
        loss = float(o_o.step(lambda: F.mse_loss(m_o(xb), yb), lambda l: l.backward()).detach())
        assert abs(loss - losses_r[t]) <= 2e-6 * abs(losses_r[t]), (t, loss, losses_r[t])
        np.testing.assert_allclose(flat(p_o).numpy(), traj_r[t].numpy(), rtol=1e-5, atol=1e-6)


def test_multiswag_five_modes_thirty_samples_against_the_reference_ensemble(ref, backend):
    """BASELINE configs[4] in the small: the reference's DeepEnsemble (ensemble.py:28-44) of five SwagOptimizer members (K = 20,
    trained separately as camelyon.py:98 does) making 30 predictions, next to ours on the same seeds -- the split
    (6 per member, member 0 takes the remainder), the order of the units, every member's use of torch's random stream
    (eps_W then eps_D per sample, Q11) and the callers' reduction logsumexp(out, 0) - log(S) (camelyon.py:29-30); then a
    second call of 7 (2 + 4 x 1 ... the remainder rule again) from the state the first one left."""
    import math
    import beyond_deep_ensembles_amd as bde
    sys.path.insert(0, REF)
    try:
        import src.algos.ensemble as rens
    finally:
        sys.path.remove(REF)
    torch.set_num_threads(1)
    x, y = data(51)

    def members(side):
        out = []
        for m in range(5):
            model = mlp(60 + m, hidden=5)
            base = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4)
            kw = dict(update_interval=1, start_epoch=0, deviation_samples=20)
            opt = ref["swag"].SwagOptimizer(model.parameters(), base, **kw) if side == "ref" else \
                bde.SwagOptimizer(model.parameters(), base, _ops=backend, **kw)
            for t in range(24):                                              # 24 updates: the ring of 20 columns wraps
                xb, yb = x[(t % 3) * 8:(t % 3 + 1) * 8], y[(t % 3) * 8:(t % 3 + 1) * 8]
                opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())
            out.append((model, opt))
        return out

    def closure(model):
        with torch.no_grad():
            return F.log_softmax(model(x[:8]), dim=-1)
    theirs, ours = rens.DeepEnsemble(members("ref")), bde.DeepEnsemble(members("ours"))
    torch.manual_seed(99)
    want = theirs.predict(closure, 30)
    torch.manual_seed(99)
    got = ours.predict(closure, 30)
    assert got.shape == want.shape == (30, 8, 2)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose((torch.logsumexp(got.float(), 0) - math.log(30)).numpy(),
                               (torch.logsumexp(want.float(), 0) - math.log(30)).numpy(), rtol=1e-5, atol=1e-6)
    # a second call continues every member's sampler where the reference's second call does
    torch.manual_seed(100)
    want2 = theirs.predict(closure, 7)
    torch.manual_seed(100)
    got2 = ours.predict(closure, 7)
    np.testing.assert_allclose(got2.numpy(), want2.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("path", ["streaming", pytest.param("small_model", marks=pytest.mark.device_unverified("svgd_small"))])
@pytest.mark.parametrize("m,d,h_override", [(5, 83, None), (5, 5411, None), (8, 4099, None), (5, 83, 0.75), (8, 4099, 3.0),
                                            (2, 7, None), (1, 33, None), (16, 1000, None), (16, 1000, 11.5),
                                            (8, 273_610, None), (5, 273_610, None)])     # BASELINE configs[1]: CIFAR ResNet-20
def test_rbf_function_against_the_reference_function(ref, backend, m, d, h_override, path):
    """`rbf(particles, h_override=None)` (svgd.py:14-32), the function itself: kernel matrix and grad_kernel for particles like
    the optimizer's (a shared start plus independent re-initialisations), median bandwidth and the override (no + 1e-8 on an
    override, svgd.py:20), M = 1 (distances all zero: h = 1e-8, kernel = exp(-0 / 2e-16) = 1) and M = 16 (two Gram tiles).
    Bars: |ours - fp64| <= max(2 |reference fp32 - fp64|, floor) per DESIGN section 3."""
    from beyond_deep_ensembles_amd.svgd import rbf
    if path == "small_model" and m > 8:
        pytest.skip("the small-model kernel takes at most 8 particles")
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(1000 * m + d)
    P = torch.randn(m, d, generator=g) * 0.05 + torch.randn(1, d, generator=g) * 0.3
    k_ref, gk_ref = ref["svgd"].rbf(P, h_override)
    k64, gk64 = ref["svgd"].rbf(P.double(), h_override)
    k, gk = rbf(P, h_override, _ops=backend, _small=(path == "small_model"))
    assert k.shape == k_ref.shape and gk.shape == gk_ref.shape
    ek, ek_ref = (k.double() - k64).abs().max(), (k_ref.double() - k64).abs().max()
    eg, eg_ref = (gk.double() - gk64).abs().max(), (gk_ref.double() - gk64).abs().max()
    assert ek <= max(2 * ek_ref, 3e-6), (float(ek), float(ek_ref))
    assert eg <= max(2 * eg_ref, 3e-6 * float(gk64.abs().max())), (float(eg), float(eg_ref), float(gk64.abs().max()))


# ------------------------------------------------- checkpoints, the other way round ("... and save back") --
def _roundtrip(obj):
    import io
    b = io.BytesIO()
    torch.save(obj, b)
    b.seek(0)
    return torch.load(b, weights_only=False)


@pytest.mark.parametrize("base_kind", ["nesterov", "adam"])
def test_the_reference_resumes_from_our_svgd_checkpoint(ref, backend, base_kind):
    """tests/test_shells.py loads checkpoints the REFERENCE wrote; this is the way back: `torch.save(opt.state_dict())` of our
    SVGDOptimizer after two steps, `load_state_dict` into the imported reference's SVGDOptimizer over a fresh model (the
    per-tensor `particle_i` entries, the string keys of svgd.py:51-56; our extra `__fused` entry is carried along unread), the
    base optimizer's state through its own state_dict -- and the reference's next two steps are our next two steps, and its
    sample_parameters() cycles through our particles from our position (svgd.py:107-112)."""
    import beyond_deep_ensembles_amd as bde
    torch.set_num_threads(1)
    x, y = data(61)
    mk = (lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4)) if base_kind == "nesterov" \
        else (lambda ps: torch.optim.Adam(ps, lr=1e-3))
    kw = dict(particle_count=5, dataset_size=50000, l2_reg=3e-4, kernel_grad_scale=1.0)
    model = mlp(3)
    torch.manual_seed(5)
    base = mk(model.parameters())
    opt = bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model), base, _ops=backend, **kw)
    for t in range(2):
        opt.step(lambda: 0.1 * F.mse_loss(model(x[:8]), y[:8]), lambda l: l.backward())
    opt.sample_parameters()                                             # position 1 of the round robin goes into the file
    ck = _roundtrip({"model": model.state_dict(), "optimizer": opt.state_dict(), "base": base.state_dict()})
    model_r = mlp(99)
    model_r.load_state_dict(ck["model"])
    base_r = mk(model_r.parameters())
    opt_r = ref["svgd"].SVGDOptimizer(model_r.parameters(), lambda: None, base_r, **kw)
    sd = ck["optimizer"]
    sd["state"]["__base_optimizer"] = base_r            # the pickled optimizer drives orphan tensors -- also in a file the reference wrote
    opt_r.load_state_dict(sd)
    base_r.load_state_dict(ck["base"])
    assert opt_r.state["__current_particle"] == 1 and opt_r.state["__particle_count"] == 5

    def rows(o, m):
        return torch.stack([flat([o.state[p][f"particle_{i}"] for p in m.parameters()]) for i in range(5)])
    np.testing.assert_array_equal(rows(opt_r, model_r).numpy(), rows(opt, model).numpy())
    for t in range(2):
        xb, yb = x[8 + 8 * t:16 + 8 * t], y[8 + 8 * t:16 + 8 * t]
        loss_r = opt_r.step(lambda: 0.1 * F.mse_loss(model_r(xb), yb), lambda l: l.backward())
        loss = opt.step(lambda: 0.1 * F.mse_loss(model(xb), yb), lambda l: l.backward())
        np.testing.assert_allclose(float(loss), float(loss_r), rtol=1e-6)
        np.testing.assert_allclose(rows(opt, model).numpy(), rows(opt_r, model_r).numpy(), rtol=1e-5, atol=1e-7)
    for _ in range(6):
        opt_r.sample_parameters()
        opt.sample_parameters()
        np.testing.assert_allclose(flat(model.parameters()).numpy(), flat(model_r.parameters()).numpy(), rtol=1e-5, atol=1e-7)


def test_the_reference_resumes_from_our_swag_and_ivon_checkpoints(ref, backend):
    """The same for SwagOptimizer (swag.py:28-35: `__mean`, `__sq_weights`, the rolled `[D, K]` `__deviations`, the three counters;
    the ring has wrapped when the file is written) and iVONOptimizer (per-parameter mean / momentum / precision, ivorn.py:29-36):
    the reference continues from our file bit for bit -- moments, deviation columns, counters, the next posterior sample on the
    same random stream; iVON's mean / precision / live parameters on the checker, within the draw's one ulp on the kernels."""
    import beyond_deep_ensembles_amd as bde
    torch.set_num_threads(1)
    x, y = data(62)
    mk = lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4)      # noqa: E731
    kw = dict(update_interval=1, start_epoch=0, deviation_samples=4)
    model = mlp(3, hidden=5)
    base = mk(model.parameters())
    opt = bde.SwagOptimizer(model.parameters(), base, _ops=backend, **kw)
    for t in range(6):
        opt.step(lambda: F.mse_loss(model(x[:8]), y[:8]), lambda l: l.backward())
    opt.complete_epoch()
    ck = _roundtrip({"model": model.state_dict(), "optimizer": opt.state_dict(), "base": base.state_dict()})
    model_r = mlp(9, hidden=5)
    model_r.load_state_dict(ck["model"])
    base_r = mk(model_r.parameters())
    opt_r = ref["swag"].SwagOptimizer(model_r.parameters(), base_r, **kw)
    sd = ck["optimizer"]
    sd["state"]["__base_optimizer"] = base_r
    opt_r.load_state_dict(sd)
    base_r.load_state_dict(ck["base"])
    for t in range(3):
        opt_r.step(lambda: F.mse_loss(model_r(x[8:16]), y[8:16]), lambda l: l.backward())
        opt.step(lambda: F.mse_loss(model(x[8:16]), y[8:16]), lambda l: l.backward())
    assert (opt_r.state["__epoch"], opt_r.state["__steps_since_swag_start"], opt_r.state["__updates"]) == \
        (opt.state["__epoch"], opt.state["__steps_since_swag_start"], opt.state["__updates"]) == (1, 9, 9)
    np.testing.assert_array_equal(opt_r.state["__mean"].numpy(), opt.mean_vector().cpu().numpy())
    np.testing.assert_array_equal(opt_r.state["__sq_weights"].numpy(), opt.sq_vector().cpu().numpy())
    np.testing.assert_array_equal(opt_r.state["__deviations"].numpy(), opt.deviations_dk().cpu().numpy())
    torch.manual_seed(1)
    opt_r.sample_parameters()
    torch.manual_seed(1)
    opt.sample_parameters()
    np.testing.assert_allclose(flat(model.parameters()).numpy(), flat(model_r.parameters()).numpy(), rtol=1e-5, atol=1e-6)

    kw = dict(lr=1e-3, prior_prec=50, dataset_size=50000, damping=1e-3, augmentation=10, mc_samples=2)
    model = mlp(3, hidden=5)
    opt = bde.iVONOptimizer(model.parameters(), _ops=backend, **kw)
    torch.manual_seed(4)
    for t in range(2):
        opt.step(lambda: F.mse_loss(model(x[:8]), y[:8]), lambda l: l.backward())
    ck = _roundtrip({"model": model.state_dict(), "optimizer": opt.state_dict()})
    model_r = mlp(9, hidden=5)
    model_r.load_state_dict(ck["model"])
    opt_r = ref["ivon"].iVONOptimizer(model_r.parameters(), **kw)
    opt_r.load_state_dict(ck["optimizer"])
    rng_state = torch.get_rng_state()
    for t in range(2):
        opt.step(lambda: F.mse_loss(model(x[8:16]), y[8:16]), lambda l: l.backward())
    torch.set_rng_state(rng_state)
    for t in range(2):
        opt_r.step(lambda: F.mse_loss(model_r(x[8:16]), y[8:16]), lambda l: l.backward())
    pr, po = list(model_r.parameters()), list(model.parameters())
    exact = type(backend).__name__ == "OracleOps"
    for key in ("mean", "momentum", "precision"):
        a, b = flat([opt_r.state[p][key] for p in pr]).numpy(), flat([opt.state[p][key] for p in po]).numpy()
        if exact:
            np.testing.assert_array_equal(b, a)
        else:
            np.testing.assert_allclose(b, a, rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(flat(po).numpy(), flat(pr).numpy(), rtol=0 if exact else 2e-6, atol=0 if exact else 2e-7)


def test_bayesian_layer_state_dicts_are_interchangeable_with_the_reference_layers():
    """`model.state_dict()` of a network of our BBBConv2d / BBBLinear layers and of the reference's (bbb_layers.py:27-41,114-131:
    `weight.mean`, `weight.rho`, `bias.mean`, `bias.rho` per layer) carry the same keys and shapes and load into each other with
    strict=True -- what `torch.save(model.state_dict())` / `DeepEnsemble.state_dict()["models"]` (ensemble.py:17-21) put on disk."""
    import beyond_deep_ensembles_amd as bde
    from tests.oracle_ops import OracleOps
    sys.path.insert(0, REF)
    try:
        import src.algos.bbb_layers as rl
        import src.algos.bbb as rb
    finally:
        sys.path.remove(REF)
    ops, rp, p = OracleOps(), rb.GaussianPrior(0, 1.0), bde.GaussianPrior(0, 1.0)
    theirs = nn.Sequential(rl.BBBConv2d(3, 4, 3, rp, rp, padding=1), nn.Flatten(), rl.BBBLinear(4 * 8 * 8, 5, rp, rp),
                           rl.BBBLinear(5, 2, rp, rp, bias=False))
    ours = nn.Sequential(bde.BBBConv2d(3, 4, 3, p, p, padding=1, _ops=ops), nn.Flatten(), bde.BBBLinear(4 * 8 * 8, 5, p, p, _ops=ops),
                         bde.BBBLinear(5, 2, p, p, bias=False, _ops=ops))
    a, b = theirs.state_dict(), ours.state_dict()
    assert list(a.keys()) == list(b.keys()) and all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
    theirs.load_state_dict(_roundtrip(b), strict=True)
    for k, v in theirs.state_dict().items():
        assert torch.equal(v, b[k])
    with torch.no_grad():
        for v in theirs.parameters():
            v.add_(0.25)
    ours.load_state_dict(_roundtrip(theirs.state_dict()), strict=True)
    for k, v in ours.state_dict().items():
        assert torch.equal(v, theirs.state_dict()[k])
