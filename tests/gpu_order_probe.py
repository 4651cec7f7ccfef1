"""TEST INFRASTRUCTURE (run by tests/test_gpu_order.py in a child process, or by hand):

    python tests/gpu_order_probe.py            # one JSON document on stdout

Collects the `-m gpu` suite exactly as the driver's `pytest tests -x -q -m gpu` does (same conftest, same ordering hook),
without running anything, and reports for every test IN COLLECTION ORDER: its node id, whether it carries
`@pytest.mark.device_unverified` (it reaches code that has not been green on an MI355X: tests/conftest.py), and which of
the explicit switches into such code its source names.  Those switches are the ONLY ways in: the library's bde_svgd_step no
longer picks the small-model kernel by itself (ABI 406), and the shells' defaults follow device_verified.py.
"""
import inspect
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# what a test has to WRITE to reach device-unverified code (kernel entry points by name, the shells' opt-in switches)
SWITCHES = ("svgd_step_small", "run_svgd_small", "sum_scalars", "mean_scalars", "conv_lrt_", "fused_conv=True", "fused_conv = True",
            'single_launch="two"', "host_fast_paths=True", "SMALL", "_small=True", "graph_replay=True", "graph_replay=graph_replay",
            "BDE_UNVERIFIED", "conv_kw", "small_kw")


class _Collected:
    def __init__(self):
        self.items = []

    def pytest_collection_finish(self, session):
        self.items = list(session.items)


def probe():
    plugin = _Collected()
    rc = pytest.main(["--collect-only", "-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests")], plugins=[plugin])
    rows = []
    for it in plugin.items:
        fn = getattr(it, "function", None)
        try:
            src = inspect.getsource(fn) if fn is not None else ""
        except (OSError, TypeError):
            src = ""
        rows.append({"id": it.nodeid, "function": it.nodeid.split("::")[-1].split("[")[0],
                     "unverified": it.get_closest_marker("device_unverified") is not None,
                     "families": sorted({a for m in it.iter_markers("device_unverified") for a in m.args}),
                     "switches": [s for s in SWITCHES if s in src]})
    return int(rc), rows


if __name__ == "__main__":
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):                    # pytest's own collection listing
        rc, rows = probe()
    json.dump({"rc": rc, "tests": rows}, sys.stdout)
