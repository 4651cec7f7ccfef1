"""The order of the `-m gpu` suite: the driver runs `pytest tests -x -q -m gpu`, so every test that only reaches
device-verified kernels must be collected AHEAD of the first test that reaches code which has never been green on an MI355X
(`@pytest.mark.device_unverified`, tests/conftest.py).  Checked on the collection itself, in a child process (no GPU needed):

  * order: no marked test ahead of an unmarked one; at least the 117 tests of the last green driver suite (GPUTEST_r03) ahead
    of the first marked one;
  * truthfulness of the marker, statically: an UNMARKED test whose source names one of the explicit switches into unverified
    code (tests/gpu_order_probe.py SWITCHES -- the only ways in since bde_svgd_step stopped choosing the small-model kernel by
    itself) is a parametrised shell test whose other variants carry the marker and whose body also runs on the CPU execution
    model, where tests/conftest.py::_kernel_reach_guard watches the kernels it really launches;
  * truthfulness, dynamically: that guard itself (it fails a CPU-model run that launches a never-run kernel unmarked) is
    exercised below on a made-up test.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def collected():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_order_probe.py")], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    doc = json.loads(p.stdout.decode())
    assert doc["rc"] == 0 and len(doc["tests"]) >= 150, (doc["rc"], len(doc["tests"]))
    return doc["tests"]


def test_verified_tests_are_collected_first(collected):
    flags = [t["unverified"] for t in collected]
    first = flags.index(True)
    assert not any(flags[:first]) and all(flags[first:]), "a device_unverified test is collected ahead of a verified one"
    assert first >= 117, f"only {first} tests ahead of the first one that reaches device-unverified code"
    # the multi-rank tests of each part come last in it (one process per rank on one device: the heaviest tests)
    verified_files = [t["id"].split("::")[0] for t in collected[:first]]
    assert verified_files.index("tests/test_ops_gpu.py") < verified_files.index("tests/test_shells.py") < verified_files.index("tests/test_dist_gpu.py")


def test_unmarked_gpu_tests_name_no_unverified_entry_point(collected):
    by_function = {}
    for t in collected:
        by_function.setdefault((t["id"].split("::")[0], t["function"]), []).append(t)
    for (path, fn), variants in by_function.items():
        if not variants[0]["switches"] or all(v["unverified"] for v in variants):
            continue
        # a function that names a switch and has unmarked variants: the switch must be confined to marked variants, and that
        # is only accepted where the CPU-model guard sees what the unmarked variants really launch
        assert any(v["unverified"] for v in variants), \
            f"{path}::{fn} names {variants[0]['switches']} but none of its variants carries @pytest.mark.device_unverified"
        assert path == "tests/test_shells.py" and any("[hip" in v["id"] for v in variants), \
            f"{path}::{fn}: mixed marked / unmarked variants are only accepted for the shell tests that also run on the CPU model"
    # every never-run kernel family of tools/kernel_table.py has marked tests waiting for the device
    families = {f for t in collected for f in t["families"]}
    assert {"svgd_small", "conv_lrt", "mean_scalars", "small_step_host", "fast_loop"} <= families, families


def test_the_reach_guard_fails_an_unmarked_cpu_model_run():
    """conftest._kernel_reach_guard on a made-up test file: an `emu`-backend test that launches sum_scalars_kernel (never run on
    an MI355X) fails unmarked and passes marked.  Run in a child pytest over a temporary file inside tests/ (so that the
    repository's conftest applies)."""
    from tests.hip_emu import build
    if not build.available():
        pytest.skip("no host clang / HIP headers to build the CPU model with")
    body = '''
import pytest, torch
from tests.hip_emu import emu_ops

@pytest.fixture(params=["emu"])
def backend(request):
    with emu_ops.emulated(emu_ops.ALL) as ops:
        yield ops, torch.device("cpu")

def _launch(ops):
    out = torch.zeros(())
    ops.mean_scalars([torch.tensor(1.0), torch.tensor(2.0)], out, 2.0)
    assert float(out) == 1.5

def test_unmarked(backend):
    _launch(backend[0])

@pytest.mark.device_unverified("mean_scalars")
def test_marked(backend):
    _launch(backend[0])

def test_verified_kernel_only(backend):
    ops = backend[0]
    m, d = 2, 64
    P = torch.randn(m, d)
    ops.svgd_gram(P, d, ops.svgd_ws(m, "cpu"))
'''
    path = os.path.join(ROOT, "tests", "test_zz_reach_guard_probe.py")
    try:
        with open(path, "w") as f:
            f.write(body)
        p = subprocess.run([sys.executable, "-m", "pytest", path, "-q", "-p", "no:cacheprovider"], cwd=ROOT, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=900)
    finally:
        os.remove(path)
    out = p.stdout.decode()
    assert "3 passed, 1 error" in out, out[-3000:]                      # (the guard fails the test in its teardown: an error)
    assert "test_unmarked" in out and "sum_scalars_kernel" in out and "device_unverified" in out, out[-3000:]
