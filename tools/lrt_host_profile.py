"""Where the host time of BBBLinear forward + backward goes (the kernels take ~40 us).  With lib/_bde_host.so the
layer runs C++ autograd nodes (forward ~18 us, forward + backward ~140 us of host time, mostly the autograd engine's
thread hand-off); BDE_NO_HOST_HELPER=1 shows the Python nodes (38 / 260 us; the per-Function split below applies to
those)."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beyond_deep_ensembles_amd as bde
from beyond_deep_ensembles_amd import bbb_layers as BL

dev = torch.device("cuda:0")
prior = bde.GaussianPrior(0, 1.0)
layer = bde.BBBLinear(2048, 182, prior, prior, rng="philox").to(dev)
x = torch.randn(16, 2048, device=dev)
xg = x.clone().requires_grad_(True)
leaves = [xg, layer.weight.mean, layer.weight.rho, layer.bias.mean, layer.bias.rho]

acc = {}
def wrap(owner, name, key):
    fn = getattr(owner, name)
    def inner(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
        return r
    setattr(owner, name, staticmethod(inner) if isinstance(owner, type) and name in ("forward", "backward") else inner)

wrap(BL._LrtLinear, "forward", "Function.forward")
wrap(BL._LrtLinear, "backward", "Function.backward")
ops = layer.weight._ops if hasattr(layer.weight, "_ops") else None

def cpu_time(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    acc.clear()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    return dt / n * 1e6

with torch.no_grad():
    print("forward only (no_grad)      us/iter (host)", round(cpu_time(lambda: layer(x)), 1), {k: round(v / 300 * 1e6, 1) for k, v in acc.items()})
print("forward (grad mode)         us/iter (host)", round(cpu_time(lambda: layer(xg)), 1), {k: round(v / 300 * 1e6, 1) for k, v in acc.items()})
print("forward + sum               us/iter (host)", round(cpu_time(lambda: layer(xg).sum()), 1))
print("forward + sum + grad        us/iter (host)", round(cpu_time(lambda: torch.autograd.grad(layer(xg).sum(), leaves)), 1), {k: round(v / 300 * 1e6, 1) for k, v in acc.items()})
out = layer(xg)
g = torch.ones_like(out)
print("forward + grad(g given)     us/iter (host)", round(cpu_time(lambda: torch.autograd.grad(layer(xg), leaves, grad_outputs=g)), 1), {k: round(v / 300 * 1e6, 1) for k, v in acc.items()})

pr = cProfile.Profile()
pr.enable()
with torch.no_grad():
    for _ in range(300):
        layer(x)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:3500])
