#!/usr/bin/env python3
"""Wide BBBLinear kernels (sigma^2 cached) across builds of the library, interleaved and repeated:
    python tools/lrt_ab.py [--shape BxIxO] [--rounds N] name=path [name=path ...]
Each (round, build) runs in its own process; per build: forward, backward without / with the input gradient (us)."""
import os, sys, subprocess
args = sys.argv[1:]
shape, rounds = "64x4096x4096", 3
while args and args[0].startswith("--"):
    if args[0] == "--shape":
        shape = args[1]
    elif args[0] == "--rounds":
        rounds = int(args[1])
    args = args[2:]
if args and "=" in args[0]:
    res = {}
    for _ in range(rounds):
        for spec in args:
            name, path = spec.split("=")
            out = subprocess.run([sys.executable, __file__, "--shape", shape, path], capture_output=True, text=True)
            line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "FAILED " + out.stderr.strip()[-300:]
            res.setdefault(name, []).append(line)
    for name, lines in res.items():
        for l in lines:
            print(f"{name:24s} {l}", flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_deep_ensembles_amd import _lib
_lib.LIB_PATH = os.path.abspath(args[0])
from beyond_deep_ensembles_amd.ops import HipOps
dev = torch.device("cuda", 0)
ops = HipOps()
b, i, o = (int(v) for v in shape.split("x"))


def ev(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) / iters * 1e3)
    return best


x = torch.randn(b, i, device=dev)
wm, wr = torch.randn(o, i, device=dev) * 0.1, torch.randn(o, i, device=dev) - 3
bm, br = torch.randn(o, device=dev) * 0.1, torch.randn(o, device=dev) - 3
out, var = torch.empty(b, o, device=dev), torch.empty(b, o, device=dev)
g = torch.randn(b, o, device=dev)
gx, gwm, gwr, gbm, gbr = torch.empty_like(x), torch.empty_like(wm), torch.empty_like(wr), torch.empty_like(bm), torch.empty_like(br)
s2, ds2 = torch.empty_like(wr), torch.empty_like(wr)
ops.lrt_sigma_cache(wr, s2, ds2)
t_f = ev(lambda: ops.lrt_linear_fwd(x, wm, wr, bm, br, True, out, var, seed=1, stream_id=2, w_s2=s2))
t_w = ev(lambda: ops.lrt_linear_bwd(x, wm, wr, br, True, g, var, None, gwm, gwr, gbm, gbr, seed=1, stream_id=2, w_s2=s2, w_ds2=ds2))
t_b = ev(lambda: ops.lrt_linear_bwd(x, wm, wr, br, True, g, var, gx, gwm, gwr, gbm, gbr, seed=1, stream_id=2, w_s2=s2, w_ds2=ds2))
t_fu = ev(lambda: ops.lrt_linear_fwd(x, wm, wr, bm, br, True, out, var, seed=1, stream_id=2))
print(f"fwd {t_f:6.1f}  bwd(no gx) {t_w:6.1f}  bwd {t_b:6.1f}  (gx part {t_b - t_w:6.1f})  fwd uncached {t_fu:6.1f}")
