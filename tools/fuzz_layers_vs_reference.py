"""Build-container tool (needs /root/reference): the Bayesian layers at random shapes next to the IMPORTED reference layers
(bbb_layers.py:61-80 BBBLinear, :146-154 BBBConv2d, sampling="activations"), ours over the kernel SOURCES on the CPU execution
model through the C++ autograd nodes -- output, input gradient and the four / two parameter gradients, same weights, same noise.

    python tools/fuzz_layers_vs_reference.py <first seed> <trials> [linear|conv|conv_fused ...]

"conv": what BBBConv2d() runs by default (stock convolutions + fused element-wise passes); "conv_fused": fused_conv=True (the
never-run convolution kernels).  Bar: |ours - fp64| <= max(3 |reference fp32 - fp64|, 5e-6 of the tensor's largest entry)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)
import src.algos.bbb_layers as rl        # noqa: E402
import src.algos.bbb as rb               # noqa: E402
sys.path.remove(REF)
import beyond_deep_ensembles_amd as bde  # noqa: E402
import beyond_deep_ensembles_amd.bbb_layers as L  # noqa: E402
from beyond_deep_ensembles_amd.ops import HipOps  # noqa: E402
from tests.hip_emu import build, emu_ops  # noqa: E402


def grads(layer, x, eps_holder, eps, gout, double=False):
    layer.train()
    xx = x.clone().double().requires_grad_(True) if double else x.clone().requires_grad_(True)
    eps_holder[0] = eps.double() if double else eps
    out = layer(xx)
    out.backward(gout.double() if double else gout)
    res = [out.detach(), xx.grad] + [p.grad for p in layer.parameters()]
    return [r.double() for r in res]


def copy_params(dst, src):
    with torch.no_grad():
        for a, b in zip(dst.parameters(), src.parameters()):
            a.copy_(b)


def trial(kind, seed, ops):
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    rp, p = rb.GaussianPrior(0, 1.0), bde.GaussianPrior(0, 1.0)
    bias = bool(rng.integers(0, 2)) if kind != "linear" else True        # the reference's BBBLinear needs its bias on this path
    if kind == "linear":
        i, o, b = int(rng.integers(1, 200)), int(rng.integers(1, 200)), int(rng.integers(1, 129))
        theirs = rl.BBBLinear(i, o, rp, rp)
        theirs64 = rl.BBBLinear(i, o, rp, rp).double()
        ours = bde.BBBLinear(i, o, p, p, rng="torch", _ops=ops)
        x = torch.randn(b, i, generator=g)
        desc = f"linear {b}x{i}->{o}"
    else:
        c, o, k = int(rng.integers(1, 20)), int(rng.integers(1, 20)), int(rng.choice([1, 3, 3, 5]))
        stride, pad = int(rng.integers(1, 3)), int(rng.integers(0, k))
        n, h, w = int(rng.integers(1, 5)), int(rng.integers(k, 13)), int(rng.integers(k, 13))
        theirs = rl.BBBConv2d(c, o, k, rp, rp, stride=stride, padding=pad, bias=bias)
        theirs64 = rl.BBBConv2d(c, o, k, rp, rp, stride=stride, padding=pad, bias=bias).double()
        ours = bde.BBBConv2d(c, o, k, p, p, stride=stride, padding=pad, bias=bias, rng="torch", _ops=ops,
                             **({"fused_conv": True} if kind == "conv_fused" else {}))
        x = torch.randn(n, c, h, w, generator=g)
        desc = f"{kind} n{n} {c}->{o} k{k} s{stride} p{pad} {h}x{w} bias={bias}"
    with torch.no_grad():
        for q in theirs.parameters():
            if getattr(q, "_is_gaussian_rho", False):
                q.copy_(torch.empty(q.shape).uniform_(-4.0, -1.0, generator=g))
            else:
                q.copy_(torch.randn(q.shape, generator=g) * 0.2)
    copy_params(theirs64, theirs)
    copy_params(ours, theirs)
    holder = [None]
    old_r, old_o = rl.normal_like, L.normal_like
    rl.normal_like = lambda t: holder[0] if holder[0] is not None else torch.zeros_like(t)
    L.normal_like = rl.normal_like
    try:
        with torch.no_grad():
            shape = theirs(x).shape
        eps, gout = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
        r32 = grads(theirs, x, holder, eps, gout)
        r64 = grads(theirs64, x, holder, eps, gout, double=True)
        o32 = grads(ours, x, holder, eps, gout)
    finally:
        rl.normal_like, L.normal_like = old_r, old_o
    worst = 0.0
    for name, a, b, c in zip(["out", "gx", "g0", "g1", "g2", "g3"], o32, r32, r64):
        e_ours, e_ref = float((a - c).abs().max()), float((b - c).abs().max())
        bar = max(3 * e_ref, 5e-6 * float(c.abs().max()))
        worst = max(worst, e_ours / bar if bar > 0 else 0.0)
        if e_ours > bar:
            return False, f"{desc}: {name} |ours-fp64| {e_ours:.3e} > bar {bar:.3e} (reference {e_ref:.3e})"
    return True, f"{desc} worst {worst:.2f} of its bar"


def main():
    first, trials = int(sys.argv[1]), int(sys.argv[2])
    kinds = sys.argv[3:] or ["linear", "conv", "conv_fused"]
    torch.set_num_threads(1)
    native = build.load_host_nodes(emu_ops.ALL)                 # the C++ autograd nodes over the CPU model, as on the device
    L._native_nodes = lambda ops: native if isinstance(ops, HipOps) else None
    count = {k: [0, 0] for k in kinds}
    worst = {k: 0.0 for k in kinds}
    with emu_ops.emulated(emu_ops.ALL) as ops:
        for seed in range(first, first + trials):
            for k in kinds:
                ok, msg = trial(k, seed * 3 + len(k), ops)
                count[k][0] += 1
                if ok:
                    worst[k] = max(worst[k], float(msg.split(" worst ")[1].split()[0]))
                if not ok:
                    count[k][1] += 1
                    print(f"seed {seed} FAIL {msg}")
    for k, (n, bad) in count.items():
        print(f"{k}: {n} trials, {bad} outside the bar; largest |ours - fp64| seen: {worst[k]:.2f} of its bar")
    print("kernels launched:", ", ".join(f"{k} x{v}" for k, v in sorted(emu_ops.launched_kernels().items()) if v))


if __name__ == "__main__":
    main()
