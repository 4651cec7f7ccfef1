#!/usr/bin/env python3
"""A/B for VERDICT r4 #6 (graph-replay the small-model SVGD step), to be run on the MI355X BEFORE the feature is built:
how much of `SVGDOptimizer.step` at CIFAR ResNet-20 size (BASELINE configs[1]: 8 particles, 273,610 parameters, nesterov
SGD) is launch cost that a hipGraph replay would remove?

  (a) the product step with null closures (what BENCH's svgd_step_cifar_resnet20_shell_fused times)
  (b) the step's device work issued directly through the C ABI, nothing else: segment-table upload (pinned -> device copy),
      bde_svgd_gather_seg, bde_svgd_step_small_sgd (two launches), bde_sum_scalars
  (c) the same sequence captured ONCE in a torch.cuda.CUDAGraph and replayed (SGD's per-step scalars are constant between
      LR-scheduler steps, so no device-resident scalars are needed for this base optimizer; `first` is False after step 1)

  (d) the product step with SVGDOptimizer(graph_replay=True) (round 5: table upload + packing + the two launches replayed
      from one hipGraph per (staging slot, step scalars); the loss sum stays a launch of its own)

(a) - (b) is the shell's Python; (b) - (c) is what graph replay can buy; (a) - (d) is what the product's option buys.  Prints microseconds per iteration (host loop of
200 iterations + one synchronize; the kernels are ~15 us, so every variant is host-bound).

    python tools/graph_small_step_ab.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import beyond_deep_ensembles_amd as bde

D, M, N_TENSORS = 273_610, 8, 96


def timed(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / iters * 1e6
        best = t if best is None else min(best, t)
    return best


def main():
    dev = torch.device("cuda", 0)
    sizes = [D // N_TENSORS] * (N_TENSORS - 1)
    sizes.append(D - sum(sizes))
    params = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.05) for s in sizes]
    base = torch.optim.SGD(params, lr=0.1, momentum=0.9, nesterov=True, weight_decay=5e-4)

    def reset():
        with torch.no_grad():
            params[-1].normal_(0, 0.05)
    opt = bde.SVGDOptimizer(params, reset, base, particle_count=M, dataset_size=50000.0, l2_reg=3e-4)
    ops = opt._ops
    zero = torch.zeros((), device=dev)
    t_a = timed(lambda: opt.step(lambda: zero, lambda loss: None))
    print(f"(a) SVGDOptimizer.step, null closures:                         {t_a:7.1f} us")

    # (b) / (c): the device work of one step, on the optimizer's own buffers
    d = opt._layout.d
    st = opt._fused_buffers(base, "sgd")
    seg = opt._seg
    losses = [zero] * M
    total = torch.empty((), device=dev)
    host_table = seg.host[0]

    def device_work():
        seg.ptrs.copy_(host_table, non_blocking=True)
        ops.svgd_gather_seg(opt._G, seg, 0, M)
        ops.svgd_step_small_sgd(opt._P, opt._G, st["buf"], d, 3e-4, 1.0, 50000.0, opt._ws, opt._kstat, 0.1, 0.9, 0.0, 5e-4,
                                True, False)
        ops.sum_scalars(losses, total)
    t_b = timed(device_work)
    print(f"(b) the same device work through the C ABI, direct launches:    {t_b:7.1f} us")

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            device_work()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph):
            device_work()
        t_c = timed(graph.replay)
        print(f"(c) the same device work as ONE hipGraph replay:                {t_c:7.1f} us")
        print(f"shell Python (a - b) {t_a - t_b:6.1f} us;   launch cost a replay removes (b - c) {t_b - t_c:6.1f} us")
    except Exception as e:
        print(f"(c) capture failed: {type(e).__name__}: {e}")
    params2 = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.05) for s in sizes]
    base2 = torch.optim.SGD(params2, lr=0.1, momentum=0.9, nesterov=True, weight_decay=5e-4)

    def reset2():
        with torch.no_grad():
            params2[-1].normal_(0, 0.05)
    try:
        opt2 = bde.SVGDOptimizer(params2, reset2, base2, particle_count=M, dataset_size=50000.0, l2_reg=3e-4, graph_replay=True)
        t_d = timed(lambda: opt2.step(lambda: zero, lambda loss: None))
        print(f"(d) SVGDOptimizer(graph_replay=True).step, null closures:       {t_d:7.1f} us   ({opt2._graph_replays} replays, "
              f"{opt2._graph_captures} recordings)")
    except Exception as e:
        print(f"(d) failed: {type(e).__name__}: {e}")


if __name__ == "__main__":
    main()
