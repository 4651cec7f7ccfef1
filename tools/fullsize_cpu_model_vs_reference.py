"""Build-container tool (needs /root/reference; ~8 GB, ~3 min; `swag` / `ivon` / `bbb` / `svgd_step nesterov|adam` as arguments: that optimizer's whole step at the same size instead): the headline workload of bench.py -- 8 particles x 23,880,950
parameters (iWildCam ResNet-50 size), SURVEY 8d's synthetic inputs (a shared backbone, the last 372,918 entries re-initialised per
particle, G ~ N(0, 0.01^2), l2_reg 0, kernel_grad_scale 1, dataset_size 129,809) -- through the kernel SOURCES on the CPU execution
model (tests/hip_emu), next to the IMPORTED reference's `rbf` (svgd.py:14-32) and the two lines that follow it in `step`
(svgd.py:86,89), each in fp32 and in fp64.  Not a measurement of anything but arithmetic."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import src.algos.svgd as rsvgd                      # noqa: E402
import src.algos.swag as rswag                      # noqa: E402
import src.algos.ivorn as rivon                     # noqa: E402
import src.algos.bbb as rbbb                        # noqa: E402
import src.algos.util as rutil                      # noqa: E402
sys.path.remove("/root/reference")
import beyond_deep_ensembles_amd as bde             # noqa: E402
from beyond_deep_ensembles_amd.svgd import rbf      # noqa: E402
from tests.hip_emu import emu_ops                   # noqa: E402


def reference_phi(P, G, l2_reg, scale, n):
    kernel, grad_kernel = rsvgd.rbf(P)                                   # svgd.py:85
    grads = G + (l2_reg / 2) * P                                         # svgd.py:86
    return kernel, grad_kernel, torch.matmul(kernel, -grads) + scale * grad_kernel / n      # svgd.py:89 (no 1 / M)


def swag(d=23_880_950, k=20, updates=25):
    """SwagOptimizer (swag.py:15-114) on one parameter vector of ResNet-50 size: SURVEY 8d's random walk (25 updates, so the
    ring of K = 20 columns has wrapped), then one posterior sample on the reference's random stream."""
    g = torch.Generator().manual_seed(1234)
    theta0 = torch.randn(d, generator=g) * 0.05
    walk = [torch.randn(d, generator=g) * 1e-3 for _ in range(updates)]

    def run(side, ops=None):
        p = torch.nn.Parameter(theta0.clone())
        base = torch.optim.SGD([p], lr=1.0)
        kw = dict(update_interval=1, start_epoch=0, deviation_samples=k)
        opt = rswag.SwagOptimizer([p], base, **kw) if side == "ref" else bde.SwagOptimizer([p], base, _ops=ops, **kw)
        for w in walk:                             # SGD with lr 1 on the gradient -w: theta_t = theta_{t-1} + w_t
            opt.step(lambda: -(p * w).sum(), lambda l: l.backward())
        torch.manual_seed(7)
        opt.sample_parameters()
        sample = p.detach().clone()
        if side == "ref":
            return opt.state["__mean"], opt.state["__sq_weights"], opt.state["__deviations"], sample, opt.state["__updates"]
        return opt.mean_vector().cpu(), opt.sq_vector().cpu(), opt.deviations_dk().cpu(), sample, opt.state["__updates"]
    t0 = time.time()
    r = run("ref")
    t1 = time.time()
    with emu_ops.emulated(emu_ops.ALL) as ops:
        o = run("ours", ops)
    print(f"SWAG, D = {d:,}, K = {k}, {updates} updates (reference {t1 - t0:.0f} s, CPU model {time.time() - t1:.0f} s): updates counter "
          f"{r[4]} / {o[4]}")
    print(f"  mean bit-exact {torch.equal(r[0], o[0])}, sq_weights bit-exact {torch.equal(r[1], o[1])}, deviations [D, K] bit-exact "
          f"{torch.equal(r[2], o[2])}")
    print(f"  posterior sample, the reference's random stream: max |ours - reference| {float((r[3] - o[3]).abs().max()):.2e} "
          f"(max |sample| {float(r[3].abs().max()):.2e}, max |sample - mean| {float((r[3] - r[0]).abs().max()):.2e})")


def ivon(d=23_880_950):
    """iVONOptimizer (ivorn.py:15-127) on one parameter vector of that size at iwildcam.yaml:199-204's values (prior_prec 100,
    damping 1e-3, augmentation 1, mc_samples 2, N = 129,809), two steps on a quadratic loss."""
    g = torch.Generator().manual_seed(99)
    theta0, target = torch.randn(d, generator=g) * 0.05, torch.randn(d, generator=g) * 0.05
    kw = dict(lr=3e-5, prior_prec=100, damping=1e-3, augmentation=1, mc_samples=2, dataset_size=129_809)

    def run(side, ops=None):
        p = torch.nn.Parameter(theta0.clone())
        opt = rivon.iVONOptimizer([p], **kw) if side == "ref" else bde.iVONOptimizer([p], _ops=ops, **kw)
        torch.manual_seed(3)
        for _ in range(2):
            opt.step(lambda: 0.5 * ((p - target) ** 2).sum(), lambda l: l.backward())
        st = opt.state[p]
        return st["mean"].detach().cpu().clone(), st["momentum"].detach().cpu().clone(), st["precision"].detach().cpu().clone(), \
            p.detach().cpu().clone()
    t0 = time.time()
    r = run("ref")
    t1 = time.time()
    with emu_ops.emulated(emu_ops.ALL) as ops:
        o = run("ours", ops)
    print(f"iVON, D = {d:,}, 2 steps x 2 MC samples (reference {t1 - t0:.0f} s, CPU model {time.time() - t1:.0f} s)")
    for name, a, b in zip(["mean", "momentum", "precision", "live parameters (mean + last draw)"], r, o):
        diff = (a - b).abs()
        print(f"  {name}: {int((a != b).sum()):,} of {d:,} entries differ, max |ours - reference| {float(diff.max()):.2e} "
              f"(max |.| {float(a.abs().max()):.2e})")


def bbb(d=23_880_950):
    """One GaussianParameter of that size (util.py:151-186, blundell_init) under BBBOptimizer (bbb.py:47-89: prior N(0, 1),
    mc_samples 2, kl_rescaling 1, N = 129,809: iwildcam.yaml:136-143) over SGD, two steps, the same noise tape on both sides: the
    draw, its backward and the fused KL value + gradients."""
    g = torch.Generator().manual_seed(5)
    mean0, target = torch.randn(d, generator=g) * 0.1, torch.randn(d, generator=g) * 0.1
    tape = [torch.randn(d, generator=g) for _ in range(4)]

    def run(side, ops=None):
        noise = list(tape)
        gp = rutil.GaussianParameter((d,)) if side == "ref" else bde.GaussianParameter((d,), _ops=ops)
        with torch.no_grad():
            gp.mean.copy_(mean0)
            gp.rho.fill_(-3.0)
        params = list(gp.parameters())
        base = torch.optim.SGD(params, lr=0.05, momentum=0.9)
        old = rutil.normal_like
        try:
            if side == "ref":
                rutil.normal_like = lambda t: noise.pop(0)
                opt = rbbb.BBBOptimizer(params, base, rbbb.GaussianPrior(0.0, 1.0), dataset_size=129_809, mc_samples=2, kl_rescaling=1.0)
            else:
                gp.noise_source = lambda rho: noise.pop(0)
                opt = bde.BBBOptimizer(params, base, bde.GaussianPrior(0.0, 1.0), dataset_size=129_809, mc_samples=2, kl_rescaling=1.0,
                                       _ops=ops)
            losses = [float(opt.step(lambda: 0.5 * ((gp.sample() - target) ** 2).sum(), lambda l: l.backward()).detach())
                      for _ in range(2)]
        finally:
            rutil.normal_like = old
        return gp.mean.detach().cpu().clone(), gp.rho.detach().cpu().clone(), losses
    t0 = time.time()
    r = run("ref")
    t1 = time.time()
    with emu_ops.emulated(emu_ops.ALL) as ops:
        o = run("ours", ops)
    print(f"BBB, one GaussianParameter of {d:,} entries, 2 steps x 2 MC samples (reference {t1 - t0:.0f} s, CPU model {time.time() - t1:.0f} s)")
    print(f"  losses reference {r[2]}  ours {o[2]}")
    for name, a, b in zip(["mean", "rho"], r[:2], o[:2]):
        print(f"  {name}: max |ours - reference| {float((a - b).abs().max()):.2e} (max |step taken| {float((a - (mean0 if name == 'mean' else -3.0)).abs().max()):.2e})")


def svgd_step(base_kind, d=23_880_950, m=8):
    """The whole SVGDOptimizer.step (svgd.py:65-105) at BASELINE configs[3]'s size in ONE process: 8 particles of one parameter
    vector of ResNet-50 size, particles 1..7 re-initialised in their last 372,918 entries only (iwildcam/models.py:118-119),
    a quadratic loss per particle, two steps; base optimizer Adam(lr 3e-5) (iwildcam.yaml:213-216) or nesterov SGD
    (cifar.yaml:218-223) -- ours on the default path (Gram -> statistics -> fused update with the shared optimizer state)."""
    g = torch.Generator().manual_seed(11)
    theta0, target = torch.randn(d, generator=g) * 0.05, torch.randn(d, generator=g) * 0.05
    heads = [(torch.rand(372_918, generator=g) * 2 - 1) / 2048 ** 0.5 for _ in range(m - 1)]
    mk = (lambda ps: torch.optim.Adam(ps, lr=3e-5, weight_decay=0)) if base_kind == "adam" else \
        (lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4))
    kw = dict(particle_count=m, dataset_size=129_809, l2_reg=0.0 if base_kind == "adam" else 3e-4, kernel_grad_scale=1.0)

    def run(side, ops=None):
        p = torch.nn.Parameter(theta0.clone())
        todo = list(heads)

        def reset():
            with torch.no_grad():
                p[-372_918:] = todo.pop(0)
        base = mk([p])
        opt = rsvgd.SVGDOptimizer([p], reset, base, **kw) if side == "ref" else bde.SVGDOptimizer([p], reset, base, _ops=ops, **kw)
        losses = []
        start = torch.stack([opt.state[p][f"particle_{i}"].detach().cpu() for i in range(m)]).clone()
        for _ in range(2):
            losses.append(float(opt.step(lambda: 0.5 * ((p - target) ** 2).sum() / 1000.0, lambda l: l.backward())))
        return torch.stack([opt.state[p][f"particle_{i}"].detach().cpu() for i in range(m)]), losses, start
    t0 = time.time()
    r, lr_, start = run("ref")
    t1 = time.time()
    with emu_ops.emulated(emu_ops.ALL) as ops:
        o, lo, start_o = run("ours", ops)
    assert torch.equal(start, start_o)
    diff = (r - o).abs()
    moved = float((r - start).abs().max())
    print(f"SVGDOptimizer.step, {m} particles x {d:,}, base {base_kind}, 2 steps (reference {t1 - t0:.0f} s, CPU model {time.time() - t1:.0f} s)")
    print(f"  returned losses reference {lr_}  ours {lo}")
    print(f"  particles: max |ours - reference| {float(diff.max()):.2e}; entries further than 1e-7 apart: {int((diff > 1e-7).sum()):,} of "
          f"{m * d:,}; largest move of an entry over the two steps {moved:.2e}")


def main():
    torch.set_num_threads(os.cpu_count())
    if sys.argv[1:2] == ["svgd_step"]:
        return svgd_step(sys.argv[2] if len(sys.argv) > 2 else "nesterov")
    if sys.argv[1:] == ["swag"]:
        return swag()
    if sys.argv[1:] == ["ivon"]:
        return ivon()
    if sys.argv[1:] == ["bbb"]:
        return bbb()
    m, d, n = 8, 23_880_950, 129_809
    g = torch.Generator().manual_seed(1234)
    P = (torch.randn(1, d, generator=g) * 0.05).repeat(m, 1)
    P[:, -372_918:] = (torch.rand(m, 372_918, generator=g) * 2 - 1) / 2048 ** 0.5
    G = torch.randn(m, d, generator=g) * 0.01
    err = lambda a, b: float((a.double() - b).abs().max())               # noqa: E731
    with emu_ops.emulated(emu_ops.ALL) as ops:
        t0 = time.time()
        k, gk = rbf(P, _ops=ops, _small=False)
        ld = (d + 63) // 64 * 64
        Pp, Gp, out = P.new_zeros((m, ld)), P.new_zeros((m, ld)), P.new_zeros((m, ld))
        Pp[:, :d], Gp[:, :d] = P, G
        ws, ks = ops.svgd_ws(m, P.device), ops.svgd_kstat(m, P.device)
        ops.svgd_step(Pp, Gp, out, d, 0.0, 1.0, float(n), -1.0, ws, ks)              # sign -1: out = -phi, what the shell hands the base optimizer
        print(f"CPU model (rbf + step): {time.time() - t0:.0f} s", flush=True)
    k32, gk32, phi32 = reference_phi(P, G, 0.0, 1.0, n)
    k64, gk64, phi64 = reference_phi(P.double(), G.double(), 0.0, 1.0, n)
    print(f"M = {m}, D = {d:,}")
    print(f"  kernel matrix  |ours - fp64| {err(k, k64):.2e}   |reference fp32 - fp64| {err(k32, k64):.2e}")
    print(f"  grad_kernel    |ours - fp64| {err(gk, gk64):.2e}   |reference fp32 - fp64| {err(gk32, gk64):.2e}   (max |.| {float(gk64.abs().max()):.2e})")
    print(f"  phi            |ours - fp64| {err(-out[:, :d], phi64):.2e}   |reference fp32 - fp64| {err(phi32, phi64):.2e}   (max |.| {float(phi64.abs().max()):.2e})")


if __name__ == "__main__":
    main()
