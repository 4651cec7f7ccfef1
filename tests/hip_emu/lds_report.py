"""TEST INFRASTRUCTURE, run by hand: LDS bank conflicts of the kernels, computed on the CPU model.

With HIP_EMU_LDS_STATS=1 the model records, for every LDS wave-instruction a kernel executes, the 64 lane addresses and
applies the CDNA4 banking rules (MI355X_MICROARCH.md, "LDS": lane groups per instruction, 32 or 64 banks, broadcast of
identical addresses) -- the figure SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE measures on the device, per source line.  The
host compiler's access widths stand in for hipcc's (a float4 access is a b128, a float a b32).

    python -m tests.hip_emu.lds_report conv            # the fused BBBConv2d kernels at the ResNet-20 layer shapes
    python -m tests.hip_emu.lds_report lrt | swag | svgd
    python -m tests.hip_emu.lds_report traffic         # the conv kernels: bytes requested per tensor, matrix work / useful work
"""
import collections
import os
import subprocess
import sys
import tempfile

os.environ["HIP_EMU_LDS_STATS"] = "1"

import torch  # noqa: E402

from tests.hip_emu import build as B  # noqa: E402
from tests.hip_emu.emu_ops import ALL, emulated  # noqa: E402

SYMBOLIZER = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"


def report(ops, title, top=14):
    lib_path = B.build(ALL)
    with tempfile.NamedTemporaryFile("r", suffix=".txt") as fh:
        ops.lib.hip_emu_lds_report(fh.name.encode())
        rows = [line.split() for line in fh.read().splitlines()]
    if not rows:
        print(f"== {title}: no LDS accesses")
        return
    out = subprocess.run([SYMBOLIZER, "--obj=" + lib_path, "--output-style=GNU", "--functions=none"] + [r[0] for r in rows],
                         stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    sites = collections.OrderedDict()
    for r, where in zip(rows, out):
        where = where.strip().replace(B.CSRC + "/", "")
        where = where.split(" (discriminator")[0]
        key = (where, int(r[1]), r[2])
        acc = sites.setdefault(key, [0, 0, 0])
        for i in range(3):
            acc[i] += int(r[3 + i])
    ideal = sum(v[1] for v in sites.values())
    cycles = sum(v[2] for v in sites.values())
    print(f"== {title}: {sum(v[0] for v in sites.values())} LDS wave-instructions, {ideal} conflict-free cycles, {cycles} with conflicts "
          f"(x{cycles / max(ideal, 1):.3f})")
    for (where, size, rw), (n, i, c) in sorted(sites.items(), key=lambda kv: -(kv[1][2] - kv[1][1]))[:top]:
        if c == i:
            continue
        print(f"   {where:38s} {'write' if rw == 'w' else 'read '} b{8 * size:<4d} {n:9d} inst  x{c / i:5.2f}  (+{c - i} cycles, "
              f"{100.0 * (c - i) / max(cycles, 1):4.1f} % of all LDS cycles)")


def conv(ops):
    layers = [("3->16 32x32", 2, 3, 32, 32, 16, 3, 1, 1), ("16->16 32x32", 2, 16, 32, 32, 16, 3, 1, 1),
              ("16->32 s2", 2, 16, 32, 32, 32, 3, 2, 1), ("32->32 16x16", 2, 32, 16, 16, 32, 3, 1, 1),
              ("32->64 s2", 2, 32, 16, 16, 64, 3, 2, 1), ("64->64 8x8", 8, 64, 8, 8, 64, 3, 1, 1),
              ("16->32 1x1 s2", 2, 16, 32, 32, 32, 1, 2, 0)]
    for name, n, c, h, w, o, k, s, p in layers:
        x = torch.randn(n, c, h, w)
        w_mu, w_rho = torch.randn(o, c, k, k) * 0.1, torch.randn(o, c, k, k) - 3
        wbuf = ops.conv_lrt_wbuf(w_mu.shape, "cpu")
        ops.conv_lrt_prep(w_mu, w_rho, wbuf)
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        out, var = torch.empty(n, o, ho, wo), torch.empty(n, o, ho, wo)
        ops.lib.hip_emu_lds_report(b"/dev/null")
        ops.conv_lrt_fwd(x, wbuf, w_mu.shape, None, False, (s, s), (p, p), out, var, seed=1, stream_id=2)
        report(ops, f"conv forward {name}")
        g = torch.randn_like(out)
        gx = torch.empty_like(x)
        ops.conv_lrt_bwd_data(g, g.clone(), wbuf, w_mu.shape, x, gx, (s, s), (p, p))
        report(ops, f"conv input gradient {name}")
        ops.conv_lrt_bwd_weight(x, g, g.clone(), w_rho, torch.empty_like(w_mu), torch.empty_like(w_mu), (s, s), (p, p))
        report(ops, f"conv weight gradient {name}")


def lrt(ops):
    for b, i, o in [(64, 1024, 1024), (16, 2048, 182)]:
        x, w_mu, w_rho = torch.randn(b, i), torch.randn(o, i) * 0.1, torch.randn(o, i) - 3
        out, var = torch.empty(b, o), torch.empty(b, o)
        ops.lib.hip_emu_lds_report(b"/dev/null")
        ops.lrt_linear_fwd(x, w_mu, w_rho, None, None, True, out, var, seed=1, stream_id=2)
        report(ops, f"BBBLinear forward {b}x{i}->{o}")
        outs = [torch.empty(b, i), torch.empty(o, i), torch.empty(o, i), None, None]
        ops.lrt_linear_bwd(x, w_mu, w_rho, None, True, torch.randn(b, o), var, *outs, seed=1, stream_id=2)
        report(ops, f"BBBLinear backward {b}x{i}->{o}")


def swag(ops):
    import tests.test_ops_gpu as G
    G.DEV = "cpu"
    ops.lib.hip_emu_lds_report(b"/dev/null")
    G.test_swag_batched_sampler_both_kernels_equal_single_samples(ops)
    report(ops, "batched SWAG sampler test (both kernels)")


def svgd(ops):
    import tests.test_ops_gpu as G
    G.DEV = "cpu"
    ops.lib.hip_emu_lds_report(b"/dev/null")
    G.test_svgd_deterministic_and_ragged_sizes(ops)
    report(ops, "SVGD streaming kernels (ragged sizes test)")


def traffic(ops):
    """Requested bytes per tensor relative to its size and matrix-instruction work relative to the layer's useful flops."""
    from tests.hip_emu.emu_ops import Traffic
    layers = [("3->16 32x32", 8, 3, 32, 32, 16, 3, 1, 1), ("16->16 32x32", 8, 16, 32, 32, 16, 3, 1, 1),
              ("16->32 s2", 8, 16, 32, 32, 32, 3, 2, 1), ("32->32 16x16", 8, 32, 16, 16, 32, 3, 1, 1),
              ("32->64 s2", 8, 32, 16, 16, 64, 3, 2, 1), ("64->64 8x8", 16, 64, 8, 8, 64, 3, 1, 1),
              ("16->32 1x1 s2", 8, 16, 32, 32, 32, 1, 2, 0)]
    mfma_flop = lambda t: t.mfma32 * 2 * 32 * 32 * 2 + t.mfma16 * 2 * 16 * 16 * 4
    for name, n, c, h, w, o, k, s, p in layers:
        x = torch.randn(n, c, h, w)
        w_mu, w_rho = torch.randn(o, c, k, k) * 0.1, torch.randn(o, c, k, k) - 3
        wbuf = ops.conv_lrt_wbuf(w_mu.shape, "cpu")
        ops.conv_lrt_prep(w_mu, w_rho, wbuf, stride=(s, s), padding=(p, p))
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        out, var = torch.empty(n, o, ho, wo), torch.empty(n, o, ho, wo)
        flop = 2 * 2 * n * o * ho * wo * c * k * k
        with Traffic(ops) as t:
            ops.conv_lrt_fwd(x, wbuf, w_mu.shape, None, False, (s, s), (p, p), out, var, seed=1, stream_id=2)
        print(f"forward         {name:14s} batch {n:2d}: x read x{t.of(x)[0] / (4 * x.numel()):5.2f}, weight buffer read {t.of(wbuf)[0] / 1e3:7.1f} KB, "
              f"out + var written x{(t.of(out)[1] + t.of(var)[1]) / (8 * out.numel()):4.2f}, matrix flop / useful flop {mfma_flop(t) / flop:5.2f}")
        g = torch.randn_like(out)
        gv, gx = g.clone(), torch.empty_like(x)
        for how, phases in (("", False), (" (per phase)", True)) if s > 1 else (("", False),):
            with Traffic(ops) as t:
                ops.conv_lrt_bwd_data(g, gv, wbuf, w_mu.shape, x, gx, (s, s), (p, p), phases=phases)
            print(f"input gradient  {name:14s} batch {n:2d}: g read x{t.of(g)[0] / (4 * g.numel()):5.2f}, weight buffer read {t.of(wbuf)[0] / 1e3:7.1f} KB, "
                  f"g_x written x{t.of(gx)[1] / (4 * gx.numel()):4.2f}, matrix flop / useful flop {mfma_flop(t) / flop:5.2f}{how}")
        gwm, gwr = torch.empty_like(w_mu), torch.empty_like(w_mu)
        with Traffic(ops) as t:
            ops.conv_lrt_bwd_weight(x, g, gv, w_rho, gwm, gwr, (s, s), (p, p))
        print(f"weight gradient {name:14s} batch {n:2d}: x read x{t.of(x)[0] / (4 * x.numel()):5.2f}, g read x{t.of(g)[0] / (4 * g.numel()):5.2f}, "
              f"all reads {t.read / 1e3:8.1f} KB, all writes {t.written / 1e3:8.1f} KB (partials + result), matrix flop / useful flop "
              f"{mfma_flop(t) / flop:5.2f}")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "conv"
    with emulated(ALL) as ops_:
        ops_.lib.hip_emu_lds_report.argtypes = [__import__("ctypes").c_char_p]
        {"conv": conv, "lrt": lrt, "swag": swag, "svgd": svgd, "traffic": traffic}[what](ops_)
