#!/usr/bin/env python3
"""Instruction mix of every kernel's hottest loop, from the compiler's gfx950 assembly (no GPU needed):

    python tools/isa_report.py                               # table on stdout
    python tools/isa_report.py --write profiles/r06_isa_report.txt

For each `__global__` kernel of beyond_deep_ensembles_amd/csrc/*.hip one representative instantiation (the one named in PICK, else
the largest) is cut into loops -- a label and a later branch back to it -- and the hot loop (the smallest one holding matrix
instructions; for kernels without any: the one with the most global memory instructions) has its instructions counted by
class: matrix (v_mfma_*), packed / scalar fp32 VALU arithmetic, lane traffic (v_readlane / v_writelane / DPP /
ds_bpermute / ds_swizzle), other VALU, LDS reads / writes, global / buffer loads and stores (with `lds` = direct-to-LDS
loads), scalar memory, waits, barriers, s_nop.  This says what a kernel's inner loop is MADE of -- whether the work is on the
matrix pipe, how many LDS reads feed a matrix instruction, whether parked scalars are read back inside the loop -- not how fast it
is; DESIGN.md section 5 quotes it next to the device measurements that exist.
"""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "beyond_deep_ensembles_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
         "--cuda-device-only", "-S"]
# preferred instantiations (substrings of the demangled name): the BASELINE operating points
PICK = {"svgd_combine_kernel": "<8, true>", "svgd_combine_seg_kernel": "<8>", "svgd_fused_kernel": "<8, 0, false, false>",
        "svgd_step_small_kernel": "<8, true, 1>", "svgd_gram_kernel": "<true>", "conv_lrt_kernel": "<16, 4, false, 0>",
        "conv_lrt_wgrad_kernel": "<16, 4>", "lrt_wide_kernel": "", "swag_sample_batched_dma_kernel": ""}

CLASSES = [("mfma", r"^v_mfma"), ("valu_fp", r"^v_(pk_)?(fma|fmac|mul|add|sub|mad|max|min|exp|log|rcp|rsq|sqrt|cvt|ldexp|frexp|fract|floor|ceil|rndne|trunc)\w*_(f32|f64|legacy_f32)"),
           ("lane", r"^(v_readlane|v_writelane|v_readfirstlane|ds_bpermute|ds_permute|ds_swizzle|v_permlane|v_mov_b32_dpp|v_\w+_dpp)"),
           ("lds_rd", r"^ds_read|^ds_load"), ("lds_wr", r"^ds_write|^ds_store"),
           ("ld_lds", r"^(global|buffer)_load_lds|^(global|buffer)_load\w*\s.*\blds\b"), ("ld", r"^(global|buffer|flat)_load"), ("st", r"^(global|buffer|flat)_(store|atomic)"),
           ("smem", r"^s_(load|buffer_load)"), ("wait", r"^s_waitcnt"), ("barrier", r"^s_barrier"), ("nop", r"^s_nop"),
           ("valu", r"^v_"), ("salu", r"^s_")]


def _asm(src):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        p = subprocess.run([HIPCC] + FLAGS + [os.path.join(CSRC, src), "-o", out], cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if p.returncode:
            raise RuntimeError(f"{src}: {p.stderr.decode()[-1500:]}")
        return open(out).read()


def _classify(ins):
    for name, pat in CLASSES:
        if re.search(pat, ins):
            return name
    return "other"


def functions(text):
    """{symbol: [lines]} for every kernel (a .amdhsa_kernel directive exists for it)."""
    kernels = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, flags=re.M))
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^(\S+):\s*(;.*)?$", line)
        if m and m.group(1) in kernels:
            cur = out.setdefault(m.group(1), [])
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            cur.append(line)
    return out


def _count(body):
    counts = {}
    for b in body:
        c = _classify(b)
        counts[c] = counts.get(c, 0) + 1
    return counts


def _instructions(lines):
    body = [ln.strip().split(";")[0].strip() for ln in lines]
    return [b for b in body if b and not b.endswith(":") and not b.startswith(".")]


def hot_loop(lines):
    """(nesting, {class: count}, instructions) of the kernel's hot loop.  A loop is the region between a label and a LATER branch
    back to it (whatever block the compiler rotated the back edge to); nesting = how many other such regions enclose it.  Kernels
    with matrix instructions: the SMALLEST region that holds any (the product loop).  Others: the region with the most global /
    buffer memory instructions (the streaming loop), the smaller one on a tie.  No backward branch (fully unrolled /
    straight-line): the whole kernel (nesting 0)."""
    pos = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            pos[m.group(1)] = i
    regions = []
    for j, ln in enumerate(lines):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)\b", ln)
        if m and m.group(1) in pos and pos[m.group(1)] < j:
            regions.append((pos[m.group(1)], j))
    if not regions:
        body = _instructions(lines)
        return 0, _count(body), len(body)
    merged = {}
    for a_, b_ in regions:                                              # several back edges to one label: one loop
        merged[a_] = max(merged.get(a_, 0), b_)
    loops = []
    for a_, b_ in merged.items():
        body = _instructions(lines[a_ + 1:b_ + 1])
        depth = 1 + sum(1 for c_, d_ in merged.items() if (c_, d_) != (a_, b_) and c_ <= a_ and d_ >= b_)
        loops.append((depth, _count(body), len(body)))
    with_mfma = [lp for lp in loops if lp[1].get("mfma", 0)]
    if with_mfma:
        return min(with_mfma, key=lambda lp: lp[2])
    mem = lambda lp: lp[1].get("ld", 0) + lp[1].get("st", 0) + lp[1].get("ld_lds", 0)     # noqa: E731
    return max(loops, key=lambda lp: (mem(lp), -lp[2]))


def report():
    sources = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and f != "version.hip")
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        texts = list(ex.map(_asm, sources))
    rows = []
    for src, text in zip(sources, texts):
        fams = {}
        for sym, lines in functions(text).items():
            dem = subprocess.run(["c++filt", sym], stdout=subprocess.PIPE).stdout.decode().strip()
            dem = re.sub(r"\(.*", "", dem).replace("void ", "").replace("bde::", "")
            fams.setdefault(dem.split("<")[0], []).append((dem, lines))
        for fam, insts in sorted(fams.items()):
            want = PICK.get(fam)
            chosen = [i for i in insts if want and want in i[0]] or insts
            dem, lines = max(chosen, key=lambda i: len(i[1]))
            depth, counts, total = hot_loop(lines)
            rows.append((src, dem, len(insts), depth, total, counts))
    return rows


def table(rows):
    cols = ["mfma", "valu_fp", "lane", "valu", "lds_rd", "lds_wr", "ld_lds", "ld", "st", "smem", "wait", "barrier", "nop", "salu"]
    head = f"{'kernel (instantiation shown)':58s} {'file':17s} {'inst':>4s} {'loop':>4s} {'ins':>5s} " + " ".join(f"{c:>7s}" for c in cols)
    out = [head]
    for src, dem, n, depth, total, counts in rows:
        out.append(f"{dem[:58]:58s} {src:17s} {n:4d} {('d' + str(depth)) if depth else 'none':>4s} {total:5d} " +
                   " ".join(f"{counts.get(c, 0):7d}" for c in cols))
    return "\n".join(out)


def main():
    text = ("# Hot-loop instruction mix per kernel from `hipcc -S --offload-arch=gfx950` (the Makefile's flags), tools/isa_report.py.\n"
            "# inst = instantiations of the template in the library; loop = dN: the hot loop (smallest backward-branch region holding matrix\n"
            "# instructions, else the region with the most global memory instructions) lies inside N - 1 other such regions; none:\n"
            "# no backward branch, whole kernel counted; ins = instructions in that loop body; then counts by class: mfma = matrix pipe,\n"
            "# valu_fp = fp32 / fp64 vector arithmetic, lane = cross-lane traffic incl. v_readlane of parked scalars, valu = every other\n"
            "# vector instruction, lds_rd / lds_wr, ld_lds = global / buffer loads straight into LDS, ld / st = global / buffer loads and\n"
            "# stores, smem = scalar loads, wait = s_waitcnt, barrier, nop = s_nop, salu = every other scalar instruction.\n"
            + table(report()) + "\n")
    if "--write" in sys.argv:
        path = sys.argv[sys.argv.index("--write") + 1]
        open(os.path.join(ROOT, path) if not os.path.isabs(path) else path, "w").write(text)
        print("wrote", path)
    else:
        print(text)


if __name__ == "__main__":
    main()
