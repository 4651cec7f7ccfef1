"""Per-call times of the fused BBBLinear forward / backward ops (kernels only, preallocated outputs) and of the
autograd path around them.  Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import beyond_deep_ensembles_amd as bde
from beyond_deep_ensembles_amd.ops import HipOps

dev = torch.device("cuda:0")
ops = HipOps()


def ev(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


cases = [(16, 2048, 182), (32, 13, 50), (64, 768, 768), (75, 2048, 1139), (64, 4096, 4096), (128, 4096, 4096)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
print(f"{'B x I x O':>18} {'fwd us':>9} {'bwd us':>9} {'bwd noX':>9} {'fwd GB/s':>9} {'bwd GB/s':>9} {'autograd f+b us':>16}"
      f" {'cache us':>9} {'fwd cached':>11} {'bwd cached':>11}")
for b, i, o in cases:
    x = torch.randn(b, i, device=dev)
    wm, wr = torch.randn(o, i, device=dev) * 0.1, torch.randn(o, i, device=dev) - 3
    bm, br = torch.randn(o, device=dev) * 0.1, torch.randn(o, device=dev) - 3
    out, var = torch.empty(b, o, device=dev), torch.empty(b, o, device=dev)
    g = torch.randn(b, o, device=dev)
    gx, gwm, gwr, gbm, gbr = torch.empty_like(x), torch.empty_like(wm), torch.empty_like(wr), torch.empty_like(bm), torch.empty_like(br)
    t_f = ev(lambda: ops.lrt_linear_fwd(x, wm, wr, bm, br, True, out, var, seed=1, stream_id=2))
    t_b = ev(lambda: ops.lrt_linear_bwd(x, wm, wr, br, True, g, var, gx, gwm, gwr, gbm, gbr, seed=1, stream_id=2))
    t_b0 = ev(lambda: ops.lrt_linear_bwd(x, wm, wr, br, True, g, var, None, gwm, gwr, gbm, gbr, seed=1, stream_id=2))
    prior = bde.GaussianPrior(0, 1.0)
    layer = bde.BBBLinear(i, o, prior, prior, rng="philox").to(dev)
    xg = x.clone().requires_grad_(True)
    leaves = [xg, layer.weight.mean, layer.weight.rho, layer.bias.mean, layer.bias.rho]
    t_a = ev(lambda: torch.autograd.grad(layer(xg).sum(), leaves), 30)
    extra = ""
    if ops.lrt_sigma_cache_wanted(i, o):           # wide layers: sigma^2 and its rho-derivative once per weight version
        s2, ds2 = torch.empty_like(wr), torch.empty_like(wr)
        t_c = ev(lambda: ops.lrt_sigma_cache(wr, s2, ds2))
        t_fc = ev(lambda: ops.lrt_linear_fwd(x, wm, wr, bm, br, True, out, var, seed=1, stream_id=2, w_s2=s2))
        t_bc = ev(lambda: ops.lrt_linear_bwd(x, wm, wr, br, True, g, var, gx, gwm, gwr, gbm, gbr, seed=1, stream_id=2,
                                             w_s2=s2, w_ds2=ds2))
        extra = f" {t_c:9.1f} {t_fc:11.1f} {t_bc:11.1f}"
    print(f"{b:>5} x{i:>5} x{o:>5} {t_f:9.1f} {t_b:9.1f} {t_b0:9.1f} {8.0 * i * o / t_f / 1e3:9.1f} {20.0 * i * o / t_b / 1e3:9.1f} {t_a:16.1f}{extra}")
