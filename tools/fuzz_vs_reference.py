"""Build-container tool (needs /root/reference): random configurations of every optimizer shell next to the IMPORTED reference,
our side over the kernel SOURCES on the CPU execution model (tests/hip_emu) -- reference -> shell -> C ABI -> kernels, no oracle.
The committed tests (tests/test_reference_operating_points.py) pin the reference's own YAML values; this walks around them.

    python tools/fuzz_vs_reference.py <first seed> <trials> [svgd|swag|ivon|bbb ...]      (FUZZ_MANY_PARTICLES=1: SVGD with 9-24 particles)

Bars: SWAG schedule / moments bit-exact; SWAG sample, SVGD particles, BBB parameters rtol 1e-5 (+ atol 1e-6, or 3e-7 of the
largest step for SVGD under a normalising base optimizer, see below); iVON rtol 2e-6 + 2e-7 absolute (mean + delta cancels in some live parameters; one ulp of sqrt in the draw: torch's MKL
sqrt is not correctly rounded).  Prints one line per failure and a summary."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)
import src.algos.svgd as rsvgd          # noqa: E402
import src.algos.swag as rswag          # noqa: E402
import src.algos.ivorn as rivon         # noqa: E402
import src.algos.bbb as rbbb            # noqa: E402
import src.algos.util as rutil          # noqa: E402
sys.path.remove(REF)
import beyond_deep_ensembles_amd as bde  # noqa: E402
from tests.hip_emu import emu_ops        # noqa: E402


def net(rng, seed):
    depth, widths = int(rng.integers(1, 4)), [6]
    for _ in range(depth):
        widths.append(int(rng.integers(1, 41)))
    widths.append(2)
    bias = bool(rng.integers(0, 2))
    torch.manual_seed(seed)
    layers = []
    for a, b in zip(widths[:-1], widths[1:]):
        layers += [nn.Linear(a, b, bias=bias), nn.Tanh()]
    return nn.Sequential(*layers[:-1])


def flat(ps):
    return torch.cat([p.detach().reshape(-1) for p in ps])


def loguni(rng, lo, hi):
    return float(np.exp(rng.uniform(np.log(lo), np.log(hi))))


def base_factory(rng):
    kind = rng.choice(["sgd", "nesterov", "adam", "adam_wd"])
    lr = loguni(rng, 1e-5, 2e-2)
    wd = loguni(rng, 1e-5, 1e-2)
    if kind == "sgd":
        return kind, lambda ps: torch.optim.SGD(ps, lr=lr, momentum=0.9)
    if kind == "nesterov":
        return kind, lambda ps: torch.optim.SGD(ps, lr=lr, momentum=0.9, nesterov=True, weight_decay=wd)
    if kind == "adam":
        return kind, lambda ps: torch.optim.Adam(ps, lr=lr)
    return kind, lambda ps: torch.optim.Adam(ps, lr=lr, weight_decay=wd)


def batches(seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(24, 6, generator=g), torch.randn(24, 2, generator=g)


def trial_svgd(rng, seed, ops):
    m = int(rng.integers(1, 9)) if os.environ.get("FUZZ_MANY_PARTICLES") is None else int(rng.integers(9, 25))
    kind, base = base_factory(rng)
    kw = dict(particle_count=m, dataset_size=float(rng.choice([10000, 50000, 129809, 269038, 302464])),
              l2_reg=float(rng.choice([0.0, 1e-5, 3e-4, 0.01])), kernel_grad_scale=float(rng.choice([1.0, 1.0, 0.5])))
    small = bool(rng.integers(0, 2)) and m <= 8          # the small-model kernel takes at most 8 particles
    x, y = batches(seed)
    res = []
    for which in ("ref", "ours"):
        model = net(np.random.default_rng(seed), seed)
        torch.manual_seed(seed + 1)
        if which == "ref":
            opt = rsvgd.SVGDOptimizer(model.parameters(), lambda: rutil.reset_model_params(model), base(model.parameters()), **kw)
        else:
            opt = bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model), base(model.parameters()), _ops=ops,
                                    **(dict(single_launch="two", host_fast_paths=True) if small else {}), **kw)
        losses, prev, step = [], None, 0.0
        for t in range(3):
            xb, yb = x[t * 8:(t + 1) * 8], y[t * 8:(t + 1) * 8]
            losses.append(float(opt.step(lambda: 0.1 * F.mse_loss(model(xb), yb), lambda l: l.backward())))
            params = list(model.parameters())
            parts = torch.stack([flat([opt.state[p][f"particle_{i}"] for p in params]) for i in range(m)])
            if prev is not None:
                step = max(step, float((parts - prev).abs().max()))
            prev = parts.clone()
        res.append((parts, losses, step))
    # Adam's first steps are +-lr whatever the size of the gradient: a component whose -phi is at rounding level may land on
    # either side in ANY two fp32 evaluations, so the bar for the normalising optimizers is a fraction of the step
    # (at most 2 lr per application, M applications per step) and only for a small share of the components
    d = (res[1][0] - res[0][0]).abs()
    tight = d > 1e-7 + 1e-5 * res[0][0].abs()
    frac_bad = float(tight.float().mean())
    ok_losses = np.allclose(res[1][1], res[0][1], rtol=1e-5)
    if kind.startswith("adam"):
        ok = ok_losses and float(d.max()) <= 2.5 * res[0][2] + 1e-7 and frac_bad <= 0.02
    else:
        ok = ok_losses and not bool(tight.any())
    return ok, f"svgd m={m} {kind} small={small} {kw} max|d|={float(d.max()):.3e} step={res[0][2]:.3e} bad={frac_bad:.4f}"


def trial_swag(rng, seed, ops):
    k = int(rng.integers(2, 31))
    kw = dict(update_interval=float(rng.choice([1, 2, 2.5, 3, 3.7])), start_epoch=int(rng.integers(0, 3)), deviation_samples=k)
    _, base = base_factory(rng)
    x, y = batches(seed)
    outs = []
    for which in ("ref", "ours"):
        model = net(np.random.default_rng(seed), seed)
        opt = rswag.SwagOptimizer(model.parameters(), base(model.parameters()), **kw) if which == "ref" else \
            bde.SwagOptimizer(model.parameters(), base(model.parameters()), _ops=ops, **kw)
        for epoch in range(4):
            for t in range(int(3 + seed % 4)):
                xb, yb = x[(t % 3) * 8:(t % 3 + 1) * 8], y[(t % 3) * 8:(t % 3 + 1) * 8]
                opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())
            opt.complete_epoch()
        counters = (opt.state["__epoch"], opt.state["__steps_since_swag_start"], opt.state["__updates"])
        sample = None
        if counters[2] > 0:
            torch.manual_seed(seed + 7)
            opt.sample_parameters()
            sample = flat(list(model.parameters()))
            opt.step(lambda: F.mse_loss(model(x[:8]), y[:8]), lambda l: l.backward())
        after = flat(list(model.parameters()))
        if which == "ref":
            stats = (opt.state["__mean"], opt.state["__sq_weights"], opt.state["__deviations"])
        else:
            stats = (opt.mean_vector(), opt.sq_vector(), opt.deviations_dk())
        outs.append(([s.cpu() for s in stats], counters, sample, after))
    ok = outs[0][1] == outs[1][1] and all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0])) \
        and torch.equal(outs[0][3], outs[1][3])
    if outs[0][2] is not None:
        ok = ok and np.allclose(outs[1][2].numpy(), outs[0][2].numpy(), rtol=1e-5, atol=1e-6)
    return ok, f"swag {kw} counters={outs[0][1]} / {outs[1][1]}"


def trial_ivon(rng, seed, ops):
    kw = dict(lr=loguni(rng, 1e-5, 1e-2), prior_prec=loguni(rng, 1.0, 500.0), dataset_size=int(rng.choice([455, 50000, 129809, 302464])),
              damping=float(rng.choice([0.0, 1e-3])), augmentation=float(rng.choice([1, 10])), mc_samples=int(rng.choice([1, 2, 5])),
              tempering=float(rng.choice([1.0, 1.0, 0.5])))
    x, y = batches(seed)
    outs = []
    for which in ("ref", "ours"):
        model = net(np.random.default_rng(seed), seed)
        opt = rivon.iVONOptimizer(model.parameters(), **kw) if which == "ref" else bde.iVONOptimizer(model.parameters(), _ops=ops, **kw)
        torch.manual_seed(seed + 3)
        losses = []
        for t in range(3):
            xb, yb = x[t * 8:(t + 1) * 8], y[t * 8:(t + 1) * 8]
            losses.append(float(opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())))
        params = list(model.parameters())
        outs.append((flat([opt.state[p]["mean"] for p in params]), flat([opt.state[p]["precision"] for p in params]),
                     flat(params), losses))
    exact = all(torch.equal(a, b) for a, b in zip(outs[0][:3], outs[1][:3]))
    ok = all(np.allclose(b.numpy(), a.numpy(), rtol=2e-6, atol=2e-7) for a, b in zip(outs[0][:3], outs[1][:3])) \
        and np.allclose(outs[1][3], outs[0][3], rtol=1e-6)
    return ok, f"ivon {kw} exact={exact}", exact


def trial_bbb(rng, seed, ops):
    mc = int(rng.choice([1, 2, 5]))
    kw = dict(dataset_size=int(rng.choice([455, 50000, 129809])), mc_samples=mc, kl_rescaling=float(rng.choice([0.2, 0.5, 1.0])),
              l2_scale=float(rng.choice([0.0, 0.0, 0.3])))
    prior_std = float(rng.choice([0.1, 1.0, 10.0]))
    _, base = base_factory(rng)
    h = int(rng.integers(1, 30))
    g = torch.Generator().manual_seed(seed)
    tape = [torch.randn(s, generator=g) for _ in range(3 * mc) for s in ((h, 6), (h,), (2, h), (2,))]
    x, y = batches(seed)

    def build(side):
        noise = [t.clone() for t in tape]
        GP = rutil.GaussianParameter if side == "ref" else (lambda size: bde.GaussianParameter(size, _ops=ops))

        class Lin(nn.Module):
            def __init__(self, i, o):
                super().__init__()
                self.weight, self.bias = GP((o, i)), GP((o,))

            def forward(self, inp):
                return F.linear(inp, self.weight.sample(), self.bias.sample())
        model = nn.Sequential(Lin(6, h), nn.Tanh(), Lin(h, 2))
        with torch.no_grad():
            for p in model.parameters():
                if getattr(p, "_is_gaussian_rho", False):
                    p.fill_(-3.0)
                else:
                    p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(seed + p.numel())) * 0.1)
        extra = nn.Parameter(torch.full((3,), 0.2))
        params = list(model.parameters()) + [extra]
        if side == "ref":
            rutil.normal_like = lambda t: noise.pop(0)
            opt = rbbb.BBBOptimizer(params, base(params), rbbb.GaussianPrior(0.0, prior_std), **kw)
        else:
            for mod in model.modules():
                if isinstance(mod, bde.GaussianParameter):
                    mod.noise_source = lambda rho: noise.pop(0)
            opt = bde.BBBOptimizer(params, base(params), bde.GaussianPrior(0.0, prior_std), _ops=ops, **kw)
        return model, extra, params, opt
    old = rutil.normal_like
    res = []
    try:
        for side in ("ref", "ours"):
            model, extra, params, opt = build(side)
            losses = []
            for t in range(3):
                xb, yb = x[t * 8:(t + 1) * 8], y[t * 8:(t + 1) * 8]
                losses.append(float(opt.step(lambda: F.mse_loss(model(xb), yb) + 0.01 * extra.sum(), lambda l: l.backward()).detach()))
            res.append((flat(params), losses))
    finally:
        rutil.normal_like = old
    ok = np.allclose(res[1][0].numpy(), res[0][0].numpy(), rtol=1e-5, atol=1e-6) and np.allclose(res[1][1], res[0][1], rtol=2e-6)
    return ok, f"bbb {kw} prior_std={prior_std} h={h} max|d|={float((res[1][0] - res[0][0]).abs().max()):.3e}"


def main():
    first, trials = int(sys.argv[1]), int(sys.argv[2])
    kinds = sys.argv[3:] or ["svgd", "swag", "ivon", "bbb"]
    torch.set_num_threads(1)
    fn = {"svgd": trial_svgd, "swag": trial_swag, "ivon": trial_ivon, "bbb": trial_bbb}
    count = {k: [0, 0] for k in kinds}
    ivon_exact = 0
    with emu_ops.emulated(emu_ops.ALL) as ops:
        for seed in range(first, first + trials):
            for k in kinds:
                rng = np.random.default_rng(seed * 7 + len(k))
                try:
                    r = fn[k](rng, seed, ops)
                except Exception as e:                                  # a configuration one side refuses
                    print(f"seed {seed} {k}: EXCEPTION {type(e).__name__}: {e}")
                    count[k][1] += 1
                    continue
                count[k][0] += 1
                if k == "ivon":
                    ivon_exact += int(r[2])
                if not r[0]:
                    count[k][1] += 1
                    print(f"seed {seed} FAIL {r[1]}")
    for k, (n, bad) in count.items():
        print(f"{k}: {n} trials, {bad} outside the bar" + (f", {ivon_exact} bit-identical" if k == "ivon" else ""))


if __name__ == "__main__":
    main()
