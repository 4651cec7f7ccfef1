"""Build-container tool (needs /root/reference; ~8 GB, ~3 min): the headline workload of bench.py -- 8 particles x 23,880,950
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
sys.path.remove("/root/reference")
from beyond_deep_ensembles_amd.svgd import rbf      # noqa: E402
from tests.hip_emu import emu_ops                   # noqa: E402


def reference_phi(P, G, l2_reg, scale, n):
    kernel, grad_kernel = rsvgd.rbf(P)                                   # svgd.py:85
    grads = G + (l2_reg / 2) * P                                         # svgd.py:86
    return kernel, grad_kernel, torch.matmul(kernel, -grads) + scale * grad_kernel / n      # svgd.py:89 (no 1 / M)


def main():
    torch.set_num_threads(os.cpu_count())
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
