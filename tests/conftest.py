import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible, so that a
    plain `pytest tests/` also works in the CPU-only build container."""
    import torch
    # tests on the CPU execution model (tests/hip_emu): a hang there (a livelock the model does not detect) must fail the
    # test, not stall the suite -- pytest-timeout's thread method also ends a main thread that sits in a C call
    try:
        import pytest_timeout  # noqa: F401
        for item in items:
            if "test_hip_emu" in item.nodeid or "[emu" in item.nodeid or "cpu_model" in item.nodeid:
                item.add_marker(pytest.mark.timeout(1500, method="thread"))
    except ImportError:
        pass
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The several-ranks-on-one-device harness (tests/spawn_one_device.py) re-runs its ranks once on the HIP runtime's
    HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION queue abort; every such event is reported here so that it is visible in the
    tail of the test log (and in gpurun_out/one_device_retries.log on the GPU box)."""
    from tests import spawn_one_device
    terminalreporter.write_line(f"one-device harness: {len(spawn_one_device.RETRIES)} rank re-run(s) on "
                                f"{spawn_one_device.FAULT_TEXT}")
    for when, nprocs, line in spawn_one_device.RETRIES:
        terminalreporter.write_line(f"  {when} nprocs={nprocs} {line}")
