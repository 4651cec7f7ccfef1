"""HOST time of SVGDOptimizer.step with NULL closures (what BENCH's svgd_step_cifar_resnet20_shell_fused times) over the stub
library in a GPU-less container: the default constructor (round 6: streaming kernels, torch's loss adds, begin / end particle
loop -- device-verified code only) against the small-model kernel + the native host paths (`single_launch="two",
host_fast_paths=True`: what the defaults become once device_verified.json holds records).  Container numbers on a stub: they say
what the gates cost in host time, nothing about the device.

    python tools/shell_null_step_cpu.py [--profile]
"""
import cProfile
import os
import pstats
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import torch

import beyond_deep_ensembles_amd as bde
import shell_host_cpu as S

ops = S.stub_ops()
torch.manual_seed(0)
shapes = S.resnet20_shapes()


def run(kind, prof=False, **kw):
    params = [torch.nn.Parameter(torch.randn(sh) * 0.05) for sh in shapes]
    base = torch.optim.SGD(params, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=3e-4) if kind == "sgd" \
        else torch.optim.Adam(params, lr=1e-3)
    opt = bde.SVGDOptimizer(params, lambda: None, base, particle_count=8, dataset_size=50000, _ops=ops, **kw)
    loss = torch.zeros(())
    fwd, bwd = (lambda: loss), (lambda l: None)
    for _ in range(50):
        opt.step(fwd, bwd)
    best = 1e9
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(500):
            opt.step(fwd, bwd)
        best = min(best, (time.perf_counter() - t0) / 500 * 1e6)
    print(f"{kind:5s} {'default constructor' if not kw else 'small-model kernel + native host paths':40s} null-closure step host: {best:6.1f} us")
    if prof:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(500):
            opt.step(fwd, bwd)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    small = dict(single_launch="two", host_fast_paths=True)
    for kind in ("sgd", "adam"):
        run(kind, prof="--profile" in sys.argv and kind == "sgd")
        run(kind, **small)
