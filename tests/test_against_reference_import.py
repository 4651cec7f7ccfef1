"""Differential tests against the IMPORTED reference (build container only: skipped when
/root/reference is absent, e.g. on the GPU box).  The shells run with the oracle checker backend on
CPU, the reference optimizers run as they are; both see the same seeds, data and noise.  This
complements the committed golden fixtures with randomized configurations."""
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
        import src.algos.bbb as bbb
        import src.algos.util as util
        import src.algos.ensemble as ens
    finally:
        sys.path.remove(REF)
    return {"svgd": svgd, "swag": swag, "ivon": ivon, "bbb": bbb, "util": util, "ens": ens}


def mlp(seed, hidden):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(6, hidden), nn.Tanh(), nn.Linear(hidden, 2))


def flat(ps):
    return torch.cat([p.detach().reshape(-1) for p in ps])


@pytest.mark.parametrize("seed,m,opt_kind,l2,scale", [(1, 2, "sgd", 0.0, 1.0), (2, 5, "adam", 1e-5, 1.0),
                                                     (3, 7, "sgd_nesterov", 0.01, 0.5), (4, 16, "adam", 0.0, 1.0),
                                                     (5, 20, "sgd", 1e-3, 1.0)])
def test_svgd_matches_imported_reference(ref, seed, m, opt_kind, l2, scale):
    import beyond_deep_ensembles_amd as bde
    from tests.oracle_ops import OracleOps
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(seed)
    x, y = torch.randn(24, 6, generator=g), torch.randn(24, 2, generator=g)

    def make_opt(ps):
        if opt_kind == "sgd":
            return torch.optim.SGD(ps, lr=0.05)
        if opt_kind == "sgd_nesterov":
            return torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-3)
        return torch.optim.Adam(ps, lr=2e-3)
    results = []
    for which in ("ref", "ours"):
        model = mlp(seed, 9)
        torch.manual_seed(100 + seed)      # the reset closure consumes the same RNG stream in both runs
        if which == "ref":
            opt = ref["svgd"].SVGDOptimizer(model.parameters(), lambda: ref["util"].reset_model_params(model),
                                            make_opt(model.parameters()), particle_count=m, dataset_size=24, l2_reg=l2,
                                            kernel_grad_scale=scale)
        else:
            opt = bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model),
                                    make_opt(model.parameters()), particle_count=m, dataset_size=24, l2_reg=l2,
                                    kernel_grad_scale=scale, _ops=OracleOps())
        losses = []
        for t in range(3):
            xb, yb = x[t * 8:(t + 1) * 8], y[t * 8:(t + 1) * 8]
            losses.append(float(opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())))
        params = list(model.parameters())
        parts = torch.stack([flat([opt.state[p][f"particle_{i}"] for p in params]) for i in range(m)])
        results.append((parts, losses))
    np.testing.assert_allclose(results[1][0].numpy(), results[0][0].numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(results[1][1], results[0][1], rtol=1e-6)


@pytest.mark.parametrize("seed,k,interval,start_epoch", [(1, 3, 1, 0), (2, 5, 2.5, 1), (3, 2, 3, 0)])
def test_swag_matches_imported_reference(ref, seed, k, interval, start_epoch):
    import beyond_deep_ensembles_amd as bde
    from tests.oracle_ops import OracleOps
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(seed)
    x, y = torch.randn(24, 6, generator=g), torch.randn(24, 2, generator=g)
    outs = []
    for which in ("ref", "ours"):
        model = mlp(seed, 5)
        base = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
        if which == "ref":
            opt = ref["swag"].SwagOptimizer(model.parameters(), base, update_interval=interval, start_epoch=start_epoch,
                                            deviation_samples=k)
        else:
            opt = bde.SwagOptimizer(model.parameters(), base, update_interval=interval, start_epoch=start_epoch,
                                    deviation_samples=k, _ops=OracleOps())
        for epoch in range(3):
            for t in range(3):
                xb, yb = x[t * 8:(t + 1) * 8], y[t * 8:(t + 1) * 8]
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
        outs.append((stats, counters, sample, after))
    assert outs[0][1] == outs[1][1]                                   # integer schedule: exact
    for a, b in zip(outs[0][0], outs[1][0]):
        np.testing.assert_array_equal(a.numpy(), b.numpy())           # moments / deviation columns: bit-exact
    np.testing.assert_array_equal(outs[0][2].numpy(), outs[1][2].numpy())   # rng="torch": the reference's random stream
    np.testing.assert_array_equal(outs[0][3].numpy(), outs[1][3].numpy())


@pytest.mark.parametrize("seed,mc,aug", [(1, 1, 1.0), (2, 3, 5.0)])
def test_ivon_matches_imported_reference(ref, seed, mc, aug):
    import beyond_deep_ensembles_amd as bde
    from tests.oracle_ops import OracleOps
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(seed)
    x, y = torch.randn(24, 6, generator=g), torch.randn(24, 2, generator=g)
    outs = []
    for which in ("ref", "ours"):
        model = mlp(seed, 5)
        kw = dict(lr=1e-2, prior_prec=20.0, dataset_size=24, damping=1e-3, augmentation=aug, mc_samples=mc)
        opt = ref["ivon"].iVONOptimizer(model.parameters(), **kw) if which == "ref" else \
            bde.iVONOptimizer(model.parameters(), _ops=OracleOps(), **kw)
        torch.manual_seed(55)            # rng="torch": one normal_like per tensor, in parameter order, as the reference
        losses = []
        for t in range(3):
            xb, yb = x[t * 8:(t + 1) * 8], y[t * 8:(t + 1) * 8]
            losses.append(float(opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())))
        params = list(model.parameters())
        outs.append((flat([opt.state[p]["mean"] for p in params]), flat([opt.state[p]["precision"] for p in params]),
                     flat(params), losses))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        np.testing.assert_array_equal(a.numpy(), b.numpy())
    assert outs[0][3] == outs[1][3]


def test_ensemble_predict_matches_imported_reference(ref):
    import beyond_deep_ensembles_amd as bde

    class Counting:
        def __init__(self):
            self.n = 0

        def sample_parameters(self):
            self.n += 1
    for samples, members in [(11, 4), (30, 5), (2, 3)]:
        a = [(nn.Linear(1, 1), Counting()) for _ in range(members)]
        b = [(nn.Linear(1, 1), Counting()) for _ in range(members)]
        ra = ref["ens"].DeepEnsemble(a).predict(lambda m: torch.zeros(2), samples)
        rb = bde.DeepEnsemble(b).predict(lambda m: torch.zeros(2), samples)
        assert ra.shape == rb.shape and [o.n for _, o in a] == [o.n for _, o in b]


@pytest.mark.parametrize("seed,prior_kind,mc,l2_scale,base_kind,freeze", [
    (1, "gauss", 1, 0.0, "sgd", False), (2, "gauss", 3, 0.3, "adam", False), (3, "mixture", 2, 0.1, "sgd", False),
    (4, "mixture", 1, 0.0, "adam", True), (5, "gauss", 2, 1.0, "sgd_wd", True)])
def test_bbb_matches_imported_reference(ref, seed, prior_kind, mc, l2_scale, base_kind, freeze):
    """BBBOptimizer (shell + checker kernels) next to the reference's BBBOptimizer over weight-sampling layers built on
    each side's GaussianParameter, same replayed noise: 4 steps, losses and every parameter.  Covers the Gaussian
    prior (fused KL), the MixturePrior (fused mixture term), plain parameters with l2_scale, frozen parameters and
    optimizer-level zero_grad semantics (bbb.py:59-89)."""
    import beyond_deep_ensembles_amd as bde
    from beyond_deep_ensembles_amd.bbb import MixturePrior
    from tests.oracle_ops import OracleOps
    ops = OracleOps()
    g = torch.Generator().manual_seed(1000 + seed)
    tape = [torch.randn(s, generator=g) for _ in range(4 * mc) for s in ((7, 6), (7,), (2, 7), (2,))]
    x, y = torch.randn(40, 6, generator=g), torch.randn(40, 2, generator=g)

    def build(side):
        noise = [t.clone() for t in tape]
        GP = ref["util"].GaussianParameter if side == "ref" else (lambda size: bde.GaussianParameter(size, _ops=ops))

        class Lin(nn.Module):
            def __init__(self, i, o):
                super().__init__()
                self.weight, self.bias = GP((o, i)), GP((o,))

            def forward(self, inp):
                return F.linear(inp, self.weight.sample(), self.bias.sample())
        torch.manual_seed(seed)
        model = nn.Sequential(Lin(6, 7), nn.Tanh(), Lin(7, 2))
        with torch.no_grad():
            for p in model.parameters():
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(seed * 7 + p.numel())) * 0.3
                        - (2.0 if getattr(p, "_is_gaussian_rho", False) else 0.0))
        extra = nn.Parameter(torch.full((3,), 0.2))
        if freeze:
            model[0].weight.mean.requires_grad_(False)
            model[2].bias.rho.requires_grad_(False)
            extra.requires_grad_(False)
        params = list(model.parameters()) + [extra]
        base = {"sgd": lambda: torch.optim.SGD(params, lr=0.05, momentum=0.9),
                "sgd_wd": lambda: torch.optim.SGD(params, lr=0.05, momentum=0.9, weight_decay=0.05),
                "adam": lambda: torch.optim.Adam(params, lr=0.01)}[base_kind]()
        if side == "ref":
            prior = ref["bbb"].GaussianPrior(0.1, 0.7) if prior_kind == "gauss" else ref["bbb"].MixturePrior(0.3, 1.0, 0.02)
            ref["util"].normal_like = lambda t: noise.pop(0)
            opt = ref["bbb"].BBBOptimizer(params, base, prior, dataset_size=40, mc_samples=mc, kl_rescaling=0.7,
                                          l2_scale=l2_scale)
        else:
            prior = bde.GaussianPrior(0.1, 0.7) if prior_kind == "gauss" else MixturePrior(0.3, 1.0, 0.02)
            for mod in model.modules():
                if isinstance(mod, bde.GaussianParameter):
                    mod.noise_source = lambda rho: noise.pop(0)
            opt = bde.BBBOptimizer(params, base, prior, dataset_size=40, mc_samples=mc, kl_rescaling=0.7,
                                   l2_scale=l2_scale, _ops=ops)
        return model, extra, params, opt

    old = ref["util"].normal_like
    try:
        m_r, e_r, p_r, o_r = build("ref")
        losses_r, traj_r = [], []
        for t in range(4):
            xb, yb = x[t * 10:(t + 1) * 10], y[t * 10:(t + 1) * 10]
            losses_r.append(float(o_r.step(lambda: F.mse_loss(m_r(xb), yb) + e_r.sum() * 0.01, lambda l: l.backward()).detach()))
            traj_r.append(flat(p_r))
    finally:
        ref["util"].normal_like = old
    m_o, e_o, p_o, o_o = build("ours")
    for t in range(4):
        xb, yb = x[t * 10:(t + 1) * 10], y[t * 10:(t + 1) * 10]
        loss = float(o_o.step(lambda: F.mse_loss(m_o(xb), yb) + e_o.sum() * 0.01, lambda l: l.backward()).detach())
        assert abs(loss - losses_r[t]) <= 2e-5 * abs(losses_r[t]), (t, loss, losses_r[t])
        np.testing.assert_allclose(flat(p_o).numpy(), traj_r[t].numpy(), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("bias,mc", [(True, 3), (False, 1)])
def test_bbb_linear_parameter_sampling_matches_imported_reference(ref, bias, mc):
    """BBBLinear(sampling="parameters") (bbb_layers.py:43-60): mc_sample weight (and bias) draws per forward, averaged --
    our layer and the reference's on the same weights and the same noise tape; plus GaussianParameter.sign_init and
    collect_kl, which the drivers of the reference call."""
    sys.path.insert(0, REF)
    try:
        import src.algos.bbb_layers as rl
    finally:
        sys.path.remove(REF)
    import beyond_deep_ensembles_amd as bde
    import beyond_deep_ensembles_amd.util as bu
    from tests.oracle_ops import OracleOps
    torch.manual_seed(5)
    rprior, prior = ref["bbb"].GaussianPrior(0, 1.0), bde.GaussianPrior(0, 1.0)
    theirs = rl.BBBLinear(6, 4, rprior, rprior, sampling="parameters", mc_sample=mc, bias=bias)
    ours = bde.BBBLinear(6, 4, prior, prior, sampling="parameters", mc_sample=mc, bias=bias, rng="torch", _ops=OracleOps())
    with torch.no_grad():
        ours.weight.mean.copy_(theirs.weight.mean)
        ours.weight.rho.copy_(theirs.weight.rho)
        if bias:
            ours.bias.mean.copy_(theirs.bias.mean)
            ours.bias.rho.copy_(theirs.bias.rho)
    x = torch.randn(5, 6)
    tape = [torch.randn(4, 6) if i % (2 if bias else 1) == 0 else torch.randn(4) for i in range(mc * (2 if bias else 1))]
    outs = []
    for mod, layer in ((ref["util"], theirs), (bu, ours)):
        t = list(tape)
        old = mod.normal_like
        mod.normal_like = lambda like: t.pop(0)
        try:
            outs.append(layer(x))
        finally:
            mod.normal_like = old
        assert not t
    torch.testing.assert_close(outs[1], outs[0], rtol=1e-6, atol=1e-7)
    torch.manual_seed(9)
    theirs.weight.sign_init()
    torch.manual_seed(9)
    ours.weight.sign_init()
    assert torch.equal(ours.weight.mean, theirs.weight.mean) and torch.equal(ours.weight.rho, theirs.weight.rho)
    net = nn.Sequential(ours, nn.Sequential(bde.BBBLinear(4, 2, prior, prior, _ops=OracleOps())))
    net.train()
    net(x)
    want = float(ours.kl) + float(net[1][0].kl)
    from beyond_deep_ensembles_amd.bbb import collect_kl                       # (bbb.py:39-43)
    assert abs(float(collect_kl(net)) - want) <= 1e-6 * abs(want)
