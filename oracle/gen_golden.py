#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Run in the build container only (it needs /root/reference, which is read-only,
hence PYTHONDONTWRITEBYTECODE):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

The fixtures are data (seeded inputs + the outputs the reference produced for
them on CPU, torch 2.10, fp32).  No reference source or bytecode is copied.
The GPU box never runs this script; tests read the committed .npz files.

Fixture list:
  svgd_phi.npz        rbf()/phi for seeded P, G            (svgd.py:14-32, 86-89)
  svgd_traj_*.npz     5-step SVGDOptimizer trajectories     (svgd.py:65-105; Q1-Q5)
  swag_schedule.npz   update-gate counter traces            (swag.py:91-97)
  swag_stats.npz      moments / deviation columns / samples (swag.py:98-114, 53-58)
  bbb.npz             GaussianParameter draw, KL, one BBBOptimizer trajectory
  bbb2.npz            BBBOptimizer with MixturePrior; with frozen parameters; over the reference's BBBConv2d CNN
  lrt.npz             the reference's BBBLinear layer: output and all five gradients at five sizes (bbb_layers.py:61-80)
  lrt_tiled.npz       the same at batch 129 / 256 / 1000 (above the fused op's 128 rows per launch)
  ivon.npz            iVON trajectories with recorded noise (ivorn.py:41-115)
  ensemble.npz        DeepEnsemble.predict sample split     (ensemble.py:37-40)
  ref_*_checkpoint.pt state_dict()s written by the reference optimizers (wire compatibility); ref_svgd_checkpoint4*:
                      four particles, nesterov SGD, two further reference steps (multi-rank resume)
  rank1.npz           BBBOptimizer over the reference's Rank1Linear, 2 components, draws recorded
  conv_lrt.npz        the reference's BBBConv2d layer: output + all five gradients at the ResNet-20 layer shapes
"""
import math
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = os.environ.get("BDE_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import src.algos.util as ref_util
import src.algos.svgd as ref_svgd
import src.algos.swag as ref_swag
import src.algos.bbb as ref_bbb
import src.algos.bbb_layers as ref_bbb_layers
import src.algos.ivorn as ref_ivon
import src.algos.ensemble as ref_ens

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(1)  # reproducible reduction order


def npy(t):
    return t.detach().cpu().numpy().copy()


def flat(ts):
    return torch.cat([t.detach().reshape(-1) for t in ts])


# ------------------------------------------------------------------ SVGD phi
def gen_svgd_phi():
    out = {}
    cases = []
    g = torch.Generator().manual_seed(1234)
    idx = 0
    for (m, d) in [(5, 751), (8, 751), (8, 2050), (3, 10), (16, 300), (2, 64), (1, 33)]:
        for l2, shared in ((0.0, False), (0.01, True)):
            if True:
                if shared:
                    # particles share a "backbone" and differ only in the last 10 % ("head")
                    theta0 = torch.randn(d, generator=g) * 0.05
                    p = theta0.repeat(m, 1)
                    head = max(1, d // 10)
                    p[:, -head:] += (torch.rand(m, head, generator=g) * 2 - 1) / math.sqrt(2048)
                else:
                    p = torch.randn(m, d, generator=g) * 0.05
                grad = torch.randn(m, d, generator=g) * 0.01
                n = 129809
                scale = 1.0 if idx % 3 else 0.5
                gv = grad.clone()
                gv += l2 / 2 * p
                kernel, grad_kernel = ref_svgd.rbf(p)
                phi = torch.matmul(kernel, -gv) + scale * grad_kernel / n
                d2 = torch.cdist(p, p, p=2) ** 2
                h = torch.sqrt(0.5 * torch.quantile(d2, 0.5) / np.log(m + 1)) + 1e-8
                out[f"P_{idx}"] = npy(p)
                out[f"G_{idx}"] = npy(grad)
                out[f"K_{idx}"] = npy(kernel)
                out[f"gradK_{idx}"] = npy(grad_kernel)
                out[f"phi_{idx}"] = npy(phi)
                out[f"h_{idx}"] = npy(h)
                # fp64 evaluation of the same formula: the tolerance anchor
                p64, g64 = p.double(), grad.double()
                gv64 = g64 + l2 / 2 * p64
                k64, gk64 = ref_svgd.rbf(p64)
                out[f"phi64_{idx}"] = npy(torch.matmul(k64, -gv64) + scale * gk64 / n)
                out[f"K64_{idx}"] = npy(k64)
                cases.append([m, d, l2, scale, n, int(shared)])
                idx += 1
    out["cases"] = np.array(cases, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "svgd_phi.npz"), **out)


# ------------------------------------------------------------------ SVGD trajectories
def make_mlp():
    return nn.Sequential(nn.Linear(13, 50), nn.ReLU(), nn.Linear(50, 1))


def gen_svgd_traj(name, make_opt, m, l2_reg, scale, steps=5):
    torch.manual_seed(7)
    model = make_mlp()
    x = torch.randn(64, 13)
    y = torch.randn(64, 1)
    base = make_opt(model.parameters())
    opt = ref_svgd.SVGDOptimizer(model.parameters(), lambda: ref_util.reset_model_params(model), base,
                                 particle_count=m, dataset_size=64, l2_reg=l2_reg, kernel_grad_scale=scale)
    params = list(model.parameters())
    init = torch.stack([flat([opt.state[p][f"particle_{i}"] for p in params]) for i in range(m)])
    traj, losses = [], []
    for t in range(steps):
        xb, yb = x[(t % 4) * 16:(t % 4 + 1) * 16], y[(t % 4) * 16:(t % 4 + 1) * 16]
        loss = opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())
        losses.append(float(loss))
        traj.append(torch.stack([flat([opt.state[p][f"particle_{i}"] for p in params]) for i in range(m)]))
    st = base.state[params[0]]
    step_count = float(st["step"]) if "step" in st else -1.0
    # which particle the model's params alias after step(): the last one (svgd.py:96)
    np.savez_compressed(os.path.join(OUT, f"svgd_traj_{name}.npz"), x=npy(x), y=npy(y), init=npy(init),
                        traj=npy(torch.stack(traj)), losses=np.array(losses, dtype=np.float64),
                        base_step_count=np.array(step_count), m=np.array(m), l2_reg=np.array(l2_reg),
                        scale=np.array(scale), model_after=npy(flat(params)))


# ------------------------------------------------------------------ SWAG
def gen_swag_schedule():
    out = {}
    cfgs = [(5, 1, 2, 3), (4, 0, 1, 3), (7, 2, 3.7, 5), (3, 0, 5, 4), (6, 3, 4, 6)]
    for ci, (steps_per_epoch, start_epoch, interval, epochs) in enumerate(cfgs):
        p = nn.Parameter(torch.zeros(3))
        base = torch.optim.SGD([p], lr=1.0)
        opt = ref_swag.SwagOptimizer([p], base, update_interval=interval, start_epoch=start_epoch, deviation_samples=4)
        trace = []
        for e in range(epochs):
            for b in range(steps_per_epoch):
                opt.step(lambda: p.sum(), lambda l: l.backward())
                trace.append([e, b, opt.state["__epoch"], opt.state["__steps_since_swag_start"], opt.state["__updates"]])
            opt.complete_epoch()
        out[f"trace_{ci}"] = np.array(trace, dtype=np.int64)
    out["cfgs"] = np.array(cfgs, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "swag_schedule.npz"), **out)


def gen_swag_stats():
    out = {}
    k = 5
    ci = 0
    cases = []
    for total_updates in (3, k, k + 7):
        for mode in ("tag", "real"):
            torch.manual_seed(11 + ci)
            p1 = nn.Parameter(torch.zeros(7) if mode == "tag" else torch.randn(7) * 0.05)
            p2 = nn.Parameter(torch.zeros(2, 3) if mode == "tag" else torch.randn(2, 3) * 0.05)
            c1 = torch.ones(7) if mode == "tag" else torch.randn(7)
            c2 = torch.ones(2, 3) if mode == "tag" else torch.randn(2, 3)
            lr = 1.0 if mode == "tag" else 0.1
            base = torch.optim.SGD([p1, p2], lr=lr)
            interval = 2
            opt = ref_swag.SwagOptimizer([p1, p2], base, update_interval=interval, start_epoch=0, deviation_samples=k)
            theta0 = flat([p1, p2])
            thetas = []
            for t in range(total_updates * interval):
                opt.step(lambda: (p1 * c1).sum() + (p2 * c2).sum(), lambda l: l.backward())
                thetas.append(flat([p1, p2]))
            assert opt.state["__updates"] == total_updates
            out[f"theta0_{ci}"] = npy(theta0)
            out[f"thetas_{ci}"] = npy(torch.stack(thetas))          # theta after every step
            out[f"mean_{ci}"] = npy(opt.state["__mean"])
            out[f"sq_{ci}"] = npy(opt.state["__sq_weights"])
            out[f"dev_{ci}"] = npy(opt.state["__deviations"])       # [D, K], newest column last
            # samples with recorded noise (eps_W first, then eps_D; swag.py:57)
            d = theta0.numel()
            samples, eps_ws, eps_ds = [], [], []
            for s in range(3):
                torch.manual_seed(100 + s)
                opt.sample_parameters()
                samples.append(flat([p1, p2]))
                torch.manual_seed(100 + s)
                eps_ws.append(torch.empty(k).normal_())
                eps_ds.append(torch.empty(d).normal_())
            out[f"samples_{ci}"] = npy(torch.stack(samples))
            out[f"eps_w_{ci}"] = npy(torch.stack(eps_ws))
            out[f"eps_d_{ci}"] = npy(torch.stack(eps_ds))
            # a step() after sampling restores the pre-sampling weights first (swag.py:38,76-82)
            opt.step(lambda: (p1 * c1).sum() + (p2 * c2).sum(), lambda l: l.backward())
            out[f"theta_after_restore_step_{ci}"] = npy(flat([p1, p2]))
            out[f"c_{ci}"] = npy(flat([c1, c2]))
            cases.append([total_updates, 1 if mode == "tag" else 0, lr, interval, k])
            ci += 1
    out["cases"] = np.array(cases, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "swag_stats.npz"), **out)


# ------------------------------------------------------------------ noise recording
class NoiseTape:
    """Replaces the reference's normal_like() so the draws can be replayed."""

    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.tape = []

    def __call__(self, tensor):
        eps = torch.empty(tensor.shape, dtype=tensor.dtype).normal_(0, 1, generator=self.gen)
        self.tape.append(eps.clone())
        return eps


# ------------------------------------------------------------------ BBB
class RefSampledLinear(nn.Module):
    """A weight-sampling linear layer over the REFERENCE's GaussianParameter
    (exercises util.py:170-171 sample() the way rank1.py:51-52 does)."""

    def __init__(self, i, o):
        super().__init__()
        self.weight = ref_util.GaussianParameter((o, i))
        self.bias = ref_util.GaussianParameter((o,))
        self.weight.blundell_init()
        self.bias.blundell_init()

    def forward(self, x):
        return F.linear(x, self.weight.sample(), self.bias.sample())


def gen_bbb():
    out = {}
    torch.manual_seed(3)
    # (a) draw + KL + grads on a raw GaussianParameter
    gp = ref_util.GaussianParameter((257,))
    gp.blundell_init()
    with torch.no_grad():
        gp.rho.copy_(torch.randn(257) * 1.5 - 2.0)
    tape = NoiseTape(5)
    old = ref_util.normal_like
    ref_util.normal_like = tape
    try:
        w = gp.sample()
        gout = torch.randn(257)
        (w * gout).sum().backward()
    finally:
        ref_util.normal_like = old
    out["a_mean"], out["a_rho"], out["a_eps"] = npy(gp.mean), npy(gp.rho), npy(tape.tape[0])
    out["a_sample"], out["a_gout"] = npy(w), npy(gout)
    out["a_gmean"], out["a_grho"] = npy(gp.mean.grad), npy(gp.rho.grad)
    for si, sigma in enumerate((0.1, 1.0, 10.0)):
        for mi, mu in enumerate((0.0, 0.3)):
            gp.mean.grad = None
            gp.rho.grad = None
            prior = ref_bbb.GaussianPrior(mu, sigma)
            kl = gp.kl_divergence(prior)
            kl.backward()
            out[f"a_kl_{si}_{mi}"] = npy(kl)
            out[f"a_kl_gmean_{si}_{mi}"] = npy(gp.mean.grad)
            out[f"a_kl_grho_{si}_{mi}"] = npy(gp.rho.grad)
    out["a_priors"] = np.array([[mu, sigma] for sigma in (0.1, 1.0, 10.0) for mu in (0.0, 0.3)])

    # (b) BBBOptimizer trajectory, UCI-housing-shaped MLP (BASELINE config #1):
    #     the reference's own BBBLinear (local reparameterisation), Adam base.
    for tag, layer_kind in (("b", "bbblinear"), ("c", "sampled")):
        torch.manual_seed(21)
        prior = ref_bbb.GaussianPrior(0, 1.0)
        if layer_kind == "bbblinear":
            model = nn.Sequential(ref_bbb_layers.BBBLinear(13, 50, prior, prior), nn.ReLU(),
                                  ref_bbb_layers.BBBLinear(50, 1, prior, prior))
        else:
            model = nn.Sequential(RefSampledLinear(13, 50), nn.ReLU(), RefSampledLinear(50, 1))
        # one plain parameter too, so the l2_scale branch (bbb.py:75-76) is exercised
        extra = nn.Parameter(torch.randn(4) * 0.1)
        x = torch.randn(48, 13)
        y = torch.randn(48, 1)
        params = list(model.parameters()) + [extra]
        base = torch.optim.Adam(params, lr=1e-2)
        opt = ref_bbb.BBBOptimizer(params, base, prior, dataset_size=48, mc_samples=2, kl_rescaling=0.5,
                                   components=1, l2_scale=0.3)
        tape = NoiseTape(9)
        old_u, old_l = ref_util.normal_like, ref_bbb_layers.normal_like
        ref_util.normal_like = tape
        ref_bbb_layers.normal_like = tape
        names = [n for n, _ in model.named_parameters()] + ["extra"]
        init = {n: npy(p) for n, p in zip(names, params)}
        losses, trajs = [], []
        try:
            for t in range(3):
                xb, yb = x[(t % 3) * 16:(t % 3 + 1) * 16], y[(t % 3) * 16:(t % 3 + 1) * 16]
                loss = opt.step(lambda: F.mse_loss(model(xb), yb) + extra.sum() * 0.01, lambda l: l.backward())
                losses.append(float(loss))
                trajs.append(flat(params))
        finally:
            ref_util.normal_like, ref_bbb_layers.normal_like = old_u, old_l
        out[f"{tag}_x"], out[f"{tag}_y"] = npy(x), npy(y)
        for n in names:
            out[f"{tag}_init/{n}"] = init[n]
        out[f"{tag}_names"] = np.array(names)
        out[f"{tag}_losses"] = np.array(losses, dtype=np.float64)
        out[f"{tag}_traj"] = npy(torch.stack(trajs))
        out[f"{tag}_n_eps"] = np.array(len(tape.tape))
        for i, e in enumerate(tape.tape):
            out[f"{tag}_eps_{i}"] = npy(e)
    np.savez_compressed(os.path.join(OUT, "bbb.npz"), **out)


def _record_bbb_run(out, tag, model, extra, opt, x, y, tape, steps=3, loss_extra=True):
    """Run `steps` BBBOptimizer steps of the reference with the noise recorded; store inputs, initial
    parameters (by name), losses, the flat parameter trajectory and the noise tape under `tag`."""
    params = list(model.parameters()) + ([extra] if extra is not None else [])
    names = [n for n, _ in model.named_parameters()] + (["extra"] if extra is not None else [])
    init = {n: npy(p) for n, p in zip(names, params)}
    old_u, old_l = ref_util.normal_like, ref_bbb_layers.normal_like
    ref_util.normal_like = tape
    ref_bbb_layers.normal_like = tape
    losses, trajs = [], []
    n_batches = x.shape[0] // 16
    try:
        for t in range(steps):
            xb, yb = x[(t % n_batches) * 16:(t % n_batches + 1) * 16], y[(t % n_batches) * 16:(t % n_batches + 1) * 16]
            if extra is not None:
                loss = opt.step(lambda: F.mse_loss(model(xb), yb) + extra.sum() * 0.01, lambda l: l.backward())
            else:
                loss = opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())
            losses.append(float(loss))
            trajs.append(flat(params))
    finally:
        ref_util.normal_like, ref_bbb_layers.normal_like = old_u, old_l
    out[f"{tag}_x"], out[f"{tag}_y"] = npy(x), npy(y)
    for n in names:
        out[f"{tag}_init/{n}"] = init[n]
    out[f"{tag}_names"] = np.array(names)
    out[f"{tag}_losses"] = np.array(losses, dtype=np.float64)
    out[f"{tag}_traj"] = npy(torch.stack(trajs))
    out[f"{tag}_n_eps"] = np.array(len(tape.tape))
    for i, e in enumerate(tape.tape):
        out[f"{tag}_eps_{i}"] = npy(e)
    return params


def gen_bbb2():
    """Round-2 additions (bbb2.npz), all produced by the reference's own classes:
      d  BBBOptimizer with a MixturePrior (bbb.py:23-37): the optimizer-level zero_grad (bbb.py:60) matters here,
         the rho gradients come only from the data term;
      e  frozen (requires_grad=False) Gaussian mean and plain parameter with weight decay on the base optimizer
         (they must not move: their .grad stays None, bbb.py:60 + torch's optimizers skip None grads);
      f  the reference's BBBConv2d + BBBLinear (bbb_layers.py:105-159, 10-90) on a small CNN, Adam."""
    out = {}
    # ---- d: MixturePrior
    torch.manual_seed(31)
    prior = ref_bbb.MixturePrior(0.5, 1.0, 0.05)
    model = nn.Sequential(RefSampledLinear(13, 20), nn.ReLU(), RefSampledLinear(20, 1))
    extra = nn.Parameter(torch.randn(4) * 0.1)
    x, y = torch.randn(48, 13), torch.randn(48, 1)
    params = list(model.parameters()) + [extra]
    base = torch.optim.SGD(params, lr=0.05, momentum=0.9)
    opt = ref_bbb.BBBOptimizer(params, base, prior, dataset_size=48, mc_samples=2, kl_rescaling=0.5, l2_scale=0.3)
    _record_bbb_run(out, "d", model, extra, opt, x, y, NoiseTape(11), steps=4)

    # ---- e: frozen parameters
    torch.manual_seed(32)
    prior = ref_bbb.GaussianPrior(0, 1.0)
    model = nn.Sequential(RefSampledLinear(13, 20), nn.ReLU(), RefSampledLinear(20, 1))
    model[0].bias.mean.requires_grad_(False)           # a frozen Gaussian mean (its rho stays trainable)
    model[2].weight.rho.requires_grad_(False)          # a frozen rho (its mean stays trainable)
    extra = nn.Parameter(torch.randn(4) * 0.1)
    frozen_plain = nn.Parameter(torch.randn(6) * 0.5, requires_grad=False)
    x, y = torch.randn(48, 13), torch.randn(48, 1)
    params = list(model.parameters()) + [extra, frozen_plain]
    base = torch.optim.SGD(params, lr=0.05, momentum=0.9, weight_decay=0.1)
    opt = ref_bbb.BBBOptimizer(params, base, prior, dataset_size=48, mc_samples=1, kl_rescaling=1.0, l2_scale=1.0)
    names = [n for n, _ in model.named_parameters()] + ["extra", "frozen_plain"]
    init = {n: npy(p) for n, p in zip(names, params)}
    tape = NoiseTape(12)
    old_u = ref_util.normal_like
    ref_util.normal_like = tape
    losses, trajs = [], []
    try:
        for t in range(3):
            xb, yb = x[t * 16:(t + 1) * 16], y[t * 16:(t + 1) * 16]
            loss = opt.step(lambda: F.mse_loss(model(xb), yb) + extra.sum() * 0.01, lambda l: l.backward())
            losses.append(float(loss))
            trajs.append(flat(params))
    finally:
        ref_util.normal_like = old_u
    out["e_x"], out["e_y"] = npy(x), npy(y)
    for n in names:
        out[f"e_init/{n}"] = init[n]
    out["e_names"] = np.array(names)
    out["e_losses"] = np.array(losses, dtype=np.float64)
    out["e_traj"] = npy(torch.stack(trajs))
    out["e_n_eps"] = np.array(len(tape.tape))
    for i, e in enumerate(tape.tape):
        out[f"e_eps_{i}"] = npy(e)

    # ---- f: BBBConv2d + BBBLinear CNN
    torch.manual_seed(33)
    prior = ref_bbb.GaussianPrior(0, 1.0)

    class RefCNN(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = ref_bbb_layers.BBBConv2d(3, 4, 3, prior, prior, padding=1)
            self.conv2 = ref_bbb_layers.BBBConv2d(4, 4, 3, prior, prior, stride=2, bias=False)
            self.fc = ref_bbb_layers.BBBLinear(4, 2, prior, prior)

        def forward(self, x):
            x = F.relu(self.conv(x))
            x = F.relu(self.conv2(x)).mean(dim=(2, 3))
            return self.fc(x)

    model = RefCNN()
    x, y = torch.randn(32, 3, 7, 7), torch.randn(32, 2)
    params = list(model.parameters())
    base = torch.optim.Adam(params, lr=1e-2)
    opt = ref_bbb.BBBOptimizer(params, base, prior, dataset_size=32, mc_samples=2, kl_rescaling=0.2)
    _record_bbb_run(out, "f", model, None, opt, x, y, NoiseTape(13), steps=3)
    np.savez_compressed(os.path.join(OUT, "bbb2.npz"), **out)


# ------------------------------------------------------------------ iVON
from lrt_cases import LRT_CASES, LRT_TILED_CASES, lrt_case_inputs   # oracle/lrt_cases.py: seeded inputs shared with the tests


def gen_lrt_tiled():
    """lrt_tiled.npz: the same record at batch sizes above 128 rows (129, 256, 1000)."""
    gen_lrt(LRT_TILED_CASES, "lrt_tiled.npz")


def gen_lrt(cases=LRT_CASES, fname="lrt.npz"):
    """lrt.npz: the REFERENCE's BBBLinear (bbb_layers.py:10-90, sampling="activations", training mode) on seeded inputs:
    the output and d(sum(out * g)) / d(input, weight mean / rho, bias mean / rho) from its autograd graph.  The [O, I]
    gradients are stored as row sums, column sums and one seeded projection (the inputs are regenerated by the tests
    from the seeds), which keeps the fixture small at the iWildCam-head size and at a wide layer."""
    out = {"cases": np.array(cases)}
    prior = ref_bbb.GaussianPrior(0, 1.0)
    for seed, b, i, o in cases:
        x, w_mu, w_rho, b_mu, b_rho, eps, g, probe = lrt_case_inputs(seed, b, i, o)
        layer = ref_bbb_layers.BBBLinear(i, o, prior, prior)
        layer.train()
        with torch.no_grad():
            layer.weight.mean.copy_(torch.from_numpy(w_mu))
            layer.weight.rho.copy_(torch.from_numpy(w_rho))
            layer.bias.mean.copy_(torch.from_numpy(b_mu))
            layer.bias.rho.copy_(torch.from_numpy(b_rho))
        xin = torch.from_numpy(x).requires_grad_(True)
        old = ref_bbb_layers.normal_like
        ref_bbb_layers.normal_like = lambda t: torch.from_numpy(eps)
        try:
            y = layer(xin)
        finally:
            ref_bbb_layers.normal_like = old
        leaves = [xin, layer.weight.mean, layer.weight.rho, layer.bias.mean, layer.bias.rho]
        gx, gwm, gwr, gbm, gbr = torch.autograd.grad(y, leaves, grad_outputs=torch.from_numpy(g))
        t = f"c{seed}_"
        out[t + "out"], out[t + "g_x"], out[t + "g_bmu"], out[t + "g_brho"] = npy(y), npy(gx), npy(gbm), npy(gbr)
        pr = torch.from_numpy(probe).double()
        for name, gw in (("g_wmu", gwm), ("g_wrho", gwr)):
            out[t + name + "_rowsum"] = gw.double().sum(1).numpy()
            out[t + name + "_colsum"] = gw.double().sum(0).numpy()
            out[t + name + "_proj"] = np.array((gw.double() * pr).sum().item())
            out[t + name + "_absmax"] = np.array(gw.abs().max().item())
    np.savez_compressed(os.path.join(OUT, fname), **out)


def gen_conv():
    """conv_lrt.npz: the REFERENCE's BBBConv2d (bbb_layers.py:105-160, sampling="activations", training mode) on seeded
    inputs: the output and d(sum(out * g)) / d(input, weight mean / rho, bias mean / rho) from its autograd graph, at the
    CIFAR ResNet-20 layer shapes (small batch) and two ragged geometries.  The bias gradients are stored in full; the
    weight gradients as per-output-channel sums, per-(c, r, q) sums, one seeded projection, the absolute maximum and a
    raw corner; the input gradient and the output as per-image-and-channel sums, per-pixel sums, a projection and a raw
    corner (the inputs are regenerated by the tests from the seeds)."""
    from conv_cases import CONV_CASES, conv_case_inputs, conv_probe_w   # oracle/conv_cases.py: seeded inputs shared with the tests
    out = {"cases": np.array(CONV_CASES)}
    prior = ref_bbb.GaussianPrior(0, 1.0)
    for seed, n, c, h, w, o, k, stride, padding, bias in CONV_CASES:
        x, w_mu, w_rho, b_mu, b_rho, eps, g, probe_x = conv_case_inputs(seed, n, c, h, w, o, k, stride, padding)
        layer = ref_bbb_layers.BBBConv2d(c, o, k, prior, prior, stride=stride, padding=padding, bias=bool(bias))
        layer.train()
        with torch.no_grad():
            layer.weight.mean.copy_(torch.from_numpy(w_mu))
            layer.weight.rho.copy_(torch.from_numpy(w_rho))
            if bias:
                layer.bias.mean.copy_(torch.from_numpy(b_mu))
                layer.bias.rho.copy_(torch.from_numpy(b_rho))
        xin = torch.from_numpy(x).requires_grad_(True)
        old = ref_bbb_layers.normal_like
        ref_bbb_layers.normal_like = lambda t: torch.from_numpy(eps)
        try:
            y = layer(xin)
        finally:
            ref_bbb_layers.normal_like = old
        leaves = [xin, layer.weight.mean, layer.weight.rho] + ([layer.bias.mean, layer.bias.rho] if bias else [])
        grads = torch.autograd.grad(y, leaves, grad_outputs=torch.from_numpy(g))
        t = f"c{seed}_"
        pw_ = torch.from_numpy(conv_probe_w(seed, o, c, k)).double()
        for name, gw in (("g_wmu", grads[1]), ("g_wrho", grads[2])):
            f64 = gw.double()
            out[t + name + "_rowsum"] = f64.sum((1, 2, 3)).numpy()         # [O]
            out[t + name + "_colsum"] = f64.sum(0).reshape(-1).numpy()     # [C K K]
            out[t + name + "_proj"] = np.array((f64 * pw_).sum().item())
            out[t + name + "_absmax"] = np.array(gw.abs().max().item())
            out[t + name + "_corner"] = npy(gw[:2, :2])                    # raw values incl. the clamped sigma^2 entries
        if bias:
            out[t + "g_bmu"], out[t + "g_brho"] = npy(grads[3]), npy(grads[4])
        pe = torch.from_numpy(eps).double()           # the output's projection uses the noise tensor: same shape, seeded
        for name, full, pr in (("out", y.detach(), pe), ("g_x", grads[0], torch.from_numpy(probe_x).double())):
            f64 = full.double()
            out[t + name + "_planes"] = f64.sum((2, 3)).numpy()          # [N, channels]
            out[t + name + "_pixels"] = f64.sum((0, 1)).numpy()          # [H, W]
            out[t + name + "_proj"] = np.array((f64 * pr).sum().item())
            out[t + name + "_absmax"] = np.array(full.abs().max().item())
            out[t + name + "_corner"] = npy(full[0, :, :2, :3])          # a few raw values incl. the clamped corner
    np.savez_compressed(os.path.join(OUT, "conv_lrt.npz"), **out)


def gen_ivon():
    out = {}
    cases = []
    for ci, (aug, mc, damping, temp) in enumerate([(1.0, 2, 1e-3, 1.0), (10.0, 2, 1e-3, 1.0), (2.0, 3, 0.0, 0.7)]):
        torch.manual_seed(31 + ci)
        model = make_mlp()
        x = torch.randn(48, 13)
        y = torch.randn(48, 1)
        params = list(model.parameters())
        opt = ref_ivon.iVONOptimizer(params, lr=1e-2, prior_prec=50.0, dataset_size=48, damping=damping,
                                     tempering=temp, augmentation=aug, mc_samples=mc)
        tape = NoiseTape(17 + ci)
        old = ref_ivon.normal_like
        ref_ivon.normal_like = tape
        init = flat(params)
        means, moms, precs, losses, after, accg, dsum = [], [], [], [], [], [], []
        try:
            for t in range(3):
                xb, yb = x[(t % 3) * 16:(t % 3 + 1) * 16], y[(t % 3) * 16:(t % 3 + 1) * 16]
                loss = opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())
                losses.append(float(loss))
                means.append(flat([opt.state[p]["mean"] for p in params]))
                moms.append(flat([opt.state[p]["momentum"] for p in params]))
                precs.append(flat([opt.state[p]["precision"] for p in params]))
                after.append(flat(params))       # params stay at the last noisy sample (Q13)
                accg.append(flat([opt.state[p]["acc_grad"] for p in params]))   # sum of the mc gradients
                dsum.append(flat([opt.state[p]["delta"] for p in params]))      # sum of the mc noise draws
            # one extra sample_parameters() as eval would call it (ivorn.py:102-115)
            opt.sample_parameters()
            out[f"eval_sample_{ci}"] = npy(flat(params))
        finally:
            ref_ivon.normal_like = old
        n_t = len(params)
        # tape holds one eps per tensor per draw; regroup into flat [n_draws, D]
        draws = [flat(tape.tape[i:i + n_t]) for i in range(0, len(tape.tape), n_t)]
        out[f"x_{ci}"], out[f"y_{ci}"], out[f"init_{ci}"] = npy(x), npy(y), npy(init)
        out[f"eps_{ci}"] = npy(torch.stack(draws))           # [3*mc + 1, D]
        out[f"means_{ci}"] = npy(torch.stack(means))
        out[f"moms_{ci}"] = npy(torch.stack(moms))
        out[f"precs_{ci}"] = npy(torch.stack(precs))
        out[f"after_{ci}"] = npy(torch.stack(after))
        out[f"acc_grad_{ci}"] = npy(torch.stack(accg))
        out[f"delta_sum_{ci}"] = npy(torch.stack(dsum))
        out[f"losses_{ci}"] = np.array(losses, dtype=np.float64)
        cases.append([aug, mc, damping, temp])
    out["cases"] = np.array(cases, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "ivon.npz"), **out)


# ------------------------------------------------------------------ ensemble split
class _CountingOpt:
    def __init__(self):
        self.n = 0

    def sample_parameters(self):
        self.n += 1


def gen_ensemble():
    rows = []
    for samples, members in [(10, 5), (30, 5), (50, 3), (7, 4), (150, 5), (3, 5), (1, 1)]:
        pairs = [(nn.Linear(1, 1), _CountingOpt()) for _ in range(members)]
        ens = ref_ens.DeepEnsemble(pairs)
        outp = ens.predict(lambda m: torch.zeros(1), samples)
        counts = [o.n for _, o in pairs]
        rows.append([samples, members, outp.shape[0]] + counts + [-1] * (5 - members))
    np.savez_compressed(os.path.join(OUT, "ensemble.npz"), rows=np.array(rows, dtype=np.int64))


# ------------------------------------------------------------------ rank-1 BBB
def gen_rank1():
    """BBBOptimizer over the reference's Rank1Linear layers (rank1.py:9-80): two components, the
    GaussianParameter.sample() draws recorded."""
    import src.algos.rank1 as ref_rank1
    out = {}
    torch.manual_seed(61)
    prior = ref_bbb.GaussianPrior(0, 1.0)
    model = nn.Sequential(ref_rank1.Rank1Linear(13, 20, prior, components=2), nn.ReLU(),
                          ref_rank1.Rank1Linear(20, 1, prior, components=2))
    x, y = torch.randn(32, 13), torch.randn(32, 1)
    params = list(model.parameters())
    names = [n for n, _ in model.named_parameters()]
    base = torch.optim.Adam(params, lr=5e-3)
    opt = ref_bbb.BBBOptimizer(params, base, prior, dataset_size=32, mc_samples=1, kl_rescaling=1.0, components=2,
                               l2_scale=1e-2)
    tape = NoiseTape(23)
    old = ref_util.normal_like
    ref_util.normal_like = tape
    for n, p in zip(names, params):
        out[f"init/{n}"] = npy(p)
    losses, traj = [], []
    try:
        for t in range(4):
            xb, yb = x[(t % 2) * 16:(t % 2 + 1) * 16], y[(t % 2) * 16:(t % 2 + 1) * 16]
            # one optimizer step = one forward per component, summed (the drivers' closure for rank-1 ensembles)
            loss = opt.step(lambda: sum(F.mse_loss(model(xb), yb) for _ in range(2)), lambda l: l.backward())
            losses.append(float(loss))
            traj.append(flat(params))
    finally:
        ref_util.normal_like = old
    out["names"] = np.array(names)
    out["x"], out["y"] = npy(x), npy(y)
    out["losses"] = np.array(losses, dtype=np.float64)
    out["traj"] = npy(torch.stack(traj))
    out["n_eps"] = np.array(len(tape.tape))
    for i, e in enumerate(tape.tape):
        out[f"eps_{i}"] = npy(e)
    np.savez_compressed(os.path.join(OUT, "rank1.npz"), **out)


# ------------------------------------------------------------------ checkpoints
def gen_checkpoints():
    """state_dict()s written by the REFERENCE optimizers (data: tensors, counters and a pickled
    torch.optim base optimizer -- no reference classes), for the wire-compatibility tests."""
    torch.manual_seed(41)
    p1, p2 = nn.Parameter(torch.randn(7) * 0.1), nn.Parameter(torch.randn(2, 3) * 0.1)
    c1, c2 = torch.randn(7), torch.randn(2, 3)
    base = torch.optim.SGD([p1, p2], lr=0.1, momentum=0.9)
    opt = ref_swag.SwagOptimizer([p1, p2], base, update_interval=2, start_epoch=0, deviation_samples=4)
    for t in range(14):
        opt.step(lambda: (p1 * c1).sum() + (p2 * c2).sum(), lambda l: l.backward())
    opt.complete_epoch()
    torch.save({"optimizer": opt.state_dict(), "params": [p1.detach().clone(), p2.detach().clone()],
                "mean": opt.state["__mean"].clone(), "sq": opt.state["__sq_weights"].clone(),
                "dev": opt.state["__deviations"].clone(), "c": [c1, c2]},
               os.path.join(OUT, "ref_swag_checkpoint.pt"))

    torch.manual_seed(42)
    model = make_mlp()
    base = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    sv = ref_svgd.SVGDOptimizer(model.parameters(), lambda: ref_util.reset_model_params(model), base,
                                particle_count=3, dataset_size=64, l2_reg=0.01)
    x, y = torch.randn(16, 13), torch.randn(16, 1)
    for t in range(2):
        sv.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward())
    params = list(model.parameters())
    particles = torch.stack([flat([sv.state[p][f"particle_{i}"] for p in params]) for i in range(3)])
    torch.save({"optimizer": sv.state_dict(), "particles": particles, "x": x, "y": y,
                "model": model.state_dict()}, os.path.join(OUT, "ref_svgd_checkpoint.pt"))
    # one more reference step from this state: what a resumed run must reproduce
    loss = sv.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward())
    after = torch.stack([flat([sv.state[p][f"particle_{i}"] for p in params]) for i in range(3)])
    torch.save({"particles_after": after, "loss": float(loss)}, os.path.join(OUT, "ref_svgd_checkpoint_next.pt"))


def gen_checkpoints4():
    """A second reference-written SVGD checkpoint with FOUR particles (divisible by 2 ranks: the multi-rank resume
    tests), SGD with momentum and nesterov + weight decay, and the reference's step after it."""
    torch.manual_seed(43)
    model = make_mlp()
    base = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4)
    sv = ref_svgd.SVGDOptimizer(model.parameters(), lambda: ref_util.reset_model_params(model), base,
                                particle_count=4, dataset_size=64, l2_reg=0.01)
    x, y = torch.randn(16, 13), torch.randn(16, 1)
    for t in range(2):
        sv.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward())
    params = list(model.parameters())
    particles = torch.stack([flat([sv.state[p][f"particle_{i}"] for p in params]) for i in range(4)])
    torch.save({"optimizer": sv.state_dict(), "particles": particles, "x": x, "y": y,
                "model": model.state_dict()}, os.path.join(OUT, "ref_svgd_checkpoint4.pt"))
    after, losses = [], []
    for t in range(2):      # two more reference steps from this state: what a resumed run must reproduce
        losses.append(float(sv.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward())))
        after.append(torch.stack([flat([sv.state[p][f"particle_{i}"] for p in params]) for i in range(4)]))
    torch.save({"particles_after": after, "losses": losses}, os.path.join(OUT, "ref_svgd_checkpoint4_next.pt"))


if __name__ == "__main__" and len(sys.argv) > 1:
    for _name in sys.argv[1:]:
        globals()["gen_" + _name]()
    print("regenerated:", sys.argv[1:])
elif __name__ == "__main__":
    gen_svgd_phi()
    gen_svgd_traj("sgd", lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4),
                  m=5, l2_reg=0.01, scale=1.0)
    gen_svgd_traj("adam", lambda ps: torch.optim.Adam(ps, lr=1e-3), m=8, l2_reg=0.0, scale=1.0)
    gen_svgd_traj("adam_wd", lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-2), m=3, l2_reg=1e-5, scale=0.5)
    gen_swag_schedule()
    gen_swag_stats()
    gen_bbb()
    gen_bbb2()
    gen_lrt()
    gen_lrt_tiled()
    gen_conv()
    gen_ivon()
    gen_ensemble()
    gen_checkpoints()
    gen_checkpoints4()
    gen_rank1()
    print("golden fixtures written to", os.path.normpath(OUT))
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f))} B")
