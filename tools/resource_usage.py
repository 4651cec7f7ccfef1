#!/usr/bin/env python3
"""Register / LDS / scratch budget of every kernel instantiation of libbde_hip.so, from the compiler's own remarks
(`hipcc -Rpass-analysis=kernel-resource-usage`, no GPU needed):

    python tools/resource_usage.py                      # table on stdout
    python tools/resource_usage.py --write profiles/r06_kernel_resource_usage.txt

One line per kernel family (template instantiations folded: count, max VGPRs / AGPRs / SGPRs, max scratch bytes per lane, max
SGPRs the allocator parked in VGPR lanes (v_writelane / v_readlane: no memory traffic, VALU slots), max VGPRs parked in AGPRs,
min occupancy in waves per SIMD, max static LDS per workgroup).  tests/test_abi.py::test_no_kernel_uses_scratch_memory asserts
the one thing that would be fatal for a streaming kernel on gfx950: no instantiation touches scratch (= HBM) for registers.
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
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
         "-Rpass-analysis=kernel-resource-usage"]                     # the Makefile's flags + the remark


def _remarks(src):
    with tempfile.TemporaryDirectory() as tmp:
        p = subprocess.run([HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, "o.o")], cwd=CSRC,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if p.returncode:
        raise RuntimeError(f"{src}: {p.stderr.decode()[-2000:]}")
    return p.stderr.decode()


def kernels(sources=None):
    """[{file, symbol, name, VGPRs, AGPRs, TotalSGPRs, ScratchSize, Occupancy, SGPRsSpill, VGPRsSpill, LDS}, ...]"""
    sources = sources or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and f != "version.hip")
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        texts = list(ex.map(_remarks, sources))
    out = []
    for src, text in zip(sources, texts):
        cur = None
        for line in text.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"file": src, "symbol": m.group(1)}
                nm = re.match(r"_ZN3bde(\d+)", cur["symbol"])
                cur["name"] = cur["symbol"][nm.end():nm.end() + int(nm.group(1))] if nm else cur["symbol"]
                out.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).replace(" ", "")] = int(m.group(2))
    return out


def table(rows):
    fams = {}
    for r in rows:
        fams.setdefault((r["file"], r["name"]), []).append(r)
    lines = [f"{'kernel':38s} {'file':18s} {'inst':>4s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'s->v':>5s} {'v->a':>5s} {'occ':>4s} {'LDS B':>7s}"]
    for (f, name), rs in sorted(fams.items()):
        mx = lambda k: max(r.get(k, 0) for r in rs)                  # noqa: E731
        lines.append(f"{name:38s} {f:18s} {len(rs):4d} {mx('VGPRs'):5d} {mx('AGPRs'):5d} {mx('TotalSGPRs'):5d} {mx('ScratchSize'):7d} "
                     f"{mx('SGPRsSpill'):5d} {mx('VGPRsSpill'):5d} {min(r.get('Occupancy', 0) for r in rs):4d} {mx('LDSSize'):7d}")
    return "\n".join(lines)


def main():
    rows = kernels()
    head = ("# hipcc -Rpass-analysis=kernel-resource-usage over beyond_deep_ensembles_amd/csrc/*.hip (gfx950, the Makefile's flags), "
            f"{len(rows)} kernel instantiations.\n# Per family: instantiations, max VGPRs / AGPRs / SGPRs, max scratch bytes per lane, "
            "s->v = max SGPRs parked in VGPR lanes (v_writelane / v_readlane),\n# v->a = max VGPRs parked in AGPRs (neither touches "
            "memory: scratch is 0 everywhere), MIN occupancy (waves per SIMD), max static LDS bytes per workgroup\n# (dynamic LDS -- "
            "the conv and batched-sampler tiles -- is chosen at launch and not in this column).  tools/resource_usage.py\n")
    text = head + table(rows) + "\n"
    if "--write" in sys.argv:
        path = sys.argv[sys.argv.index("--write") + 1]
        open(os.path.join(ROOT, path) if not os.path.isabs(path) else path, "w").write(text)
        print("wrote", path)
    else:
        print(text)


if __name__ == "__main__":
    main()
