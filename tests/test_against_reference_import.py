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
