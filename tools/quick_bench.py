#!/usr/bin/env python3
"""Per-kernel throughput probe (development tool; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_deep_ensembles_amd.ops import HipOps

D = int(os.environ.get("BDE_D", 23880950))
M, K, S = 8, 20, 30
dev = "cuda:0"
ops = HipOps()
ld = (D + 63) // 64 * 64


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def report(name, t, nbytes):
    print(f"{name:28s} {t*1e3:9.3f} ms  {nbytes/t/1e9:9.1f} GB/s  ({nbytes/t/8e12*100:5.1f}% of 8 TB/s)", flush=True)


g = torch.Generator(device=dev).manual_seed(1234)
P = torch.randn(M, ld, device=dev, generator=g) * 0.05
G = torch.randn(M, ld, device=dev, generator=g) * 0.01
out = torch.empty_like(G)
ws, ks = ops.svgd_ws(M, dev), ops.svgd_kstat(M, dev)
report("svgd_gram", timeit(lambda: ops.svgd_gram(P, D, ws)), 4 * M * D)
ops.svgd_kstats(ws, M, 0.0, 1.0, 129809.0, -1.0, ks)
report("svgd_kstats", timeit(lambda: ops.svgd_kstats(ws, M, 0.0, 1.0, 129809.0, -1.0, ks)), 1)
report("svgd_combine", timeit(lambda: ops.svgd_combine(P, G, out, D, ks)), 12 * M * D)
report("svgd_combine(inplace)", timeit(lambda: ops.svgd_combine(P, G, G, D, ks)), 12 * M * D)
report("svgd_step", timeit(lambda: ops.svgd_step(P, G, out, D, 0.0, 1.0, 129809.0, -1.0, ws, ks)), 16 * M * D)
buf = torch.zeros(ld, device=dev); ea = torch.zeros(ld, device=dev); eas = torch.zeros(ld, device=dev)
report("svgd_apply_sgd", timeit(lambda: ops.svgd_apply_sgd(P, out, buf, D, 1e-9, 0.9, 0.0, 3e-4, True, False)), (12 * M + 8) * D)
report("svgd_apply_adam", timeit(lambda: ops.svgd_apply_adam(P, out, ea, eas, D, 1e-9, 0.9, 0.999, 1e-8, 0.0, 0)), (12 * M + 16) * D)
wsn = ops.svgd_ws(M, dev)
report("svgd_fused_sgd (no gram)", timeit(lambda: ops.svgd_fused_sgd(P, G, buf, D, ks, 1e-12, 0.9, 0.0, 3e-4, True, False)), (12 * M + 8) * D)
report("svgd_fused_sgd (+next gram)", timeit(lambda: ops.svgd_fused_sgd(P, G, buf, D, ks, 1e-12, 0.9, 0.0, 3e-4, True, False, ws_next=wsn)), (12 * M + 8) * D)
report("svgd_fused_adam (+next gram)", timeit(lambda: ops.svgd_fused_adam(P, G, ea, eas, D, ks, 1e-12, 0.9, 0.999, 1e-8, 0.0, 0, ws_next=wsn)), (12 * M + 16) * D)
report("FULL STEP kstats+fused_sgd(+gram)", timeit(lambda: (ops.svgd_kstats(wsn, M, 0.0, 1.0, 129809.0, -1.0, ks), ops.svgd_fused_sgd(P, G, buf, D, ks, 1e-12, 0.9, 0.0, 3e-4, True, False, ws_next=wsn))), (12 * M + 8) * D)
del G, out

mean = torch.randn(ld, device=dev, generator=g) * 0.05
sq = mean * mean + 1e-4
devm = torch.randn(K, ld, device=dev, generator=g) * 1e-3
theta = torch.randn(ld, device=dev, generator=g) * 0.05
o = torch.empty(ld, device=dev)
report("swag_update", timeit(lambda: ops.swag_update(theta, mean, sq, devm[3], 5, D)), 24 * D)
report("swag_sample(philox)", timeit(lambda: ops.swag_sample(mean, sq, devm, 3, o, D, seed=1, stream_id=2)), 4 * D * (K + 3))
ew = torch.randn(K, device=dev); ed = torch.randn(ld, device=dev, generator=g)
report("swag_sample(eps given)", timeit(lambda: ops.swag_sample(mean, sq, devm, 3, o, D, eps_w=ew, eps_d=ed)), 4 * D * (K + 4))
ob = torch.empty(S, ld, device=dev)
t = timeit(lambda: ops.swag_sample_batched(mean, sq, devm, 3, ob, D, seed=1, stream_id0=0), iters=10)
report(f"swag_sample_batched S={S}", t, 4 * D * (K + 2 + S))
print(f"   -> {S/t:.0f} samples/s batched")
del ob, devm

rho = torch.full((ld,), -3.0, device=dev)
w = torch.empty(ld, device=dev)
report("gauss_draw_fwd(philox)", timeit(lambda: ops.gauss_draw_fwd(mean, rho, w, D, seed=1, stream_id=0)), 12 * D)
gm, gr = torch.zeros(ld, device=dev), torch.zeros(ld, device=dev)
report("gauss_draw_bwd(philox,acc)", timeit(lambda: ops.gauss_draw_bwd(w, rho, gm, gr, D, seed=1, stream_id=0, accumulate=True)), 24 * D)
rws = ops.reduce_ws(dev); kl = torch.zeros(1, device=dev)
report("gauss_kl fwd+bwd (write)", timeit(lambda: ops.gauss_kl(mean, rho, 0.0, 1.0, D, rws, kl_out=kl, gmean=gm, grho=gr)), 16 * D)
report("gauss_kl fwd+bwd (acc)", timeit(lambda: ops.gauss_kl(mean, rho, 0.0, 1.0, D, rws, kl_out=kl, gmean=gm, grho=gr, accumulate=True)), 24 * D)
var = torch.rand(ld, device=dev) + 1e-4
report("local_reparam_fwd(philox)", timeit(lambda: ops.local_reparam_fwd(mean, var, w, D, seed=1, stream_id=0)), 12 * D)
report("local_reparam_bwd(philox)", timeit(lambda: ops.local_reparam_bwd(w, var, gm, D, seed=1, stream_id=0)), 12 * D)
def torch_epi():
    e = torch.empty_like(mean).normal_()
    return mean + torch.sqrt(var) * e
report("  same epilogue, 4 torch ops", timeit(torch_epi), 12 * D)
prec = torch.full((ld,), 100.0 / 129809, device=dev); ds = torch.zeros(ld, device=dev)
report("ivon_sample(philox)", timeit(lambda: ops.ivon_sample(mean, prec, w, ds, D, 129809.0, first=False, seed=1, stream_id=0)), 20 * D)
mom = torch.zeros(ld, device=dev)
report("ivon_update", timeit(lambda: ops.ivon_update(mean, mom, prec, ds, gm, D, lam=100.0 / 129809, n_eff=129809.0, mc=2, beta1=0.9, beta2=0.999, t=1, lr=1e-9, damping=1e-3)), 32 * D)
# reference point: device-to-device copy
a = torch.empty(M * ld, device=dev); b = torch.empty(M * ld, device=dev)
report("torch copy_ (D2D)", timeit(lambda: b.copy_(a)), 8 * M * ld)
