"""TEST INFRASTRUCTURE, run by hand: which kernel template instantiations of the library do the CPU-model tests launch?

    BDE_EMU_FULL=1 python -m tests.hip_emu.coverage

Runs tests/test_hip_emu.py and the `emu` backend of tests/test_shells.py in this process, then asks the model for its launch
counts (hip_emu_launch_report) and compares them with the `*_kernel` symbols of the build.  (Kernels launched only in child
processes -- the two-rank runs of tests/test_dist_cpu.py -- are not counted.)"""
import ctypes
import re
import subprocess
import sys
import tempfile

import pytest

from tests.hip_emu import build as B
from tests.hip_emu.emu_ops import ALL


def main():
    lib_path = B.build(ALL)
    rc = pytest.main(["-q", "-x", "tests/test_hip_emu.py", "tests/test_shells.py", "-k", "emu or hip_emu", "-p", "no:cacheprovider"])
    lib = ctypes.CDLL(lib_path)
    with tempfile.NamedTemporaryFile("r", suffix=".txt") as fh:
        lib.hip_emu_launch_report(fh.name.encode())
        launched = dict(line.rsplit(" ", 1) for line in fh.read().splitlines())
    syms = subprocess.run(["nm", "-D", "--defined-only", lib_path], stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    kernels = sorted({ln.split()[-1] for ln in syms if "_kernel" in ln and ln.split()[-1].startswith("_ZN3bde")})
    miss = [k for k in kernels if k not in launched]
    print(f"pytest exit code {rc}; {len(kernels)} kernel instantiations in the library, {len(kernels) - len(miss)} launched by the CPU-model tests")
    if miss:
        names = subprocess.run(["c++filt"] + miss, stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
        families = {}
        for name in names:
            if name:
                short = re.sub(r"\(.*", "", name).replace("void ", "")
                families.setdefault(short.split("<")[0], []).append(short)
        for fam, members in sorted(families.items()):
            print(f"  never launched: {fam}: {len(members)} instantiation(s): {', '.join(m[len(fam):] or '-' for m in members[:12])}{' ...' if len(members) > 12 else ''}")
    return rc


if __name__ == "__main__":
    sys.exit(main())
