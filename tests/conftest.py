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
    config.addinivalue_line("markers", "device_unverified(*families): the test launches kernels / takes native host paths that have "
                                       "not been green on an MI355X for the sources in this tree (tools/kernel_table.py status "
                                       "'no', beyond_deep_ensembles_amd/device_verified.py); collected BEHIND every test that "
                                       "only reaches device-verified code")


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
            elif "gpu" in item.keywords:
                # a kernel that never returns (several have not run on an MI355X yet) ends the run with a stack dump after 15
                # minutes instead of holding the box until the caller's limit (the whole green suite of round 3 took under 4)
                item.add_marker(pytest.mark.timeout(900, method="thread"))
    except ImportError:
        pass
    _hardware_verified_first(items)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- order of the `-m gpu` suite: by the kernels a test can REACH ------------------------------------------
# The driver runs `pytest -x -m gpu`.  A test that launches a kernel which has never been green on an MI355X -- or takes a
# native host path written since the last device run -- carries `@pytest.mark.device_unverified` and is collected BEHIND
# every test that only reaches device-verified code, so a fault in the least verified code cannot leave the verified kernels'
# parity tests "unreached".  The marker is not a convention one has to trust:
#   * every test body that also runs on the CPU execution model (tests/test_hip_emu.py, the `emu` backend of
#     tests/test_shells.py) is watched by `_kernel_reach_guard` below: launching a kernel tools/kernel_table.py lists as
#     never-run without carrying the marker FAILS the CPU suite;
#   * tests/test_harness.py::test_unmarked_gpu_tests_name_no_unverified_entry_point reads the source of every unmarked GPU
#     test for the explicit switches that are the only way into that code (bde_svgd_step no longer picks the small-model
#     kernel by itself; the shells' defaults follow device_verified.py).
# Inside the verified part the order is round 4's: the functions of the last full green driver suite (round 3,
# GPUTEST_r03.json: 117 passed) first, verified-kernel tests written since behind them, multi-rank tests last in each part.
_NEW_SINCE_LAST_GREEN_SUITE = {
    "test_accumulating_unaligned_and_value_only_variants", "test_bbb_components_and_sample_callers_reproduce_reference_trajectory",
    "test_bbb_conv2d_fused_path_selection_and_weight_cache", "test_bbb_conv2d_layer_matches_reference_layer",
    "test_bbb_state_dict_roundtrip", "test_conv_lrt_backward", "test_conv_lrt_forward", "test_predict_distributed_rccl_one_rank",
    "test_resuming_from_a_pickled_checkpoint_continues_the_run", "test_step_hooks_and_profiler_ranges_still_work",
    "test_streaming_kernels_walk_several_grid_passes", "test_svgd_every_particle_count",
    "test_svgd_gram_load_flavour_split_does_not_change_results", "test_svgd_rccl_one_rank_forced_exchange",
    "test_svgd_small_model_fused_step", "test_svgd_small_model_kernel", "test_svgd_small_model_kernel_repeated_calls_and_rbf",
    "test_svgd_streaming_path_through_the_shell", "test_swag_batched_sampler_both_kernels_equal_single_samples",
    "test_swag_sampler_noise_statistics_over_2_to_30_normals",
}
_FILE_ORDER = ["test_abi", "test_oracle_golden", "test_philox", "test_ops_gpu", "test_fullsize_gpu", "test_shells"]
_LAST_FILES = ["test_dist_gpu", "test_dist_fullsize_gpu"]


def _order_key(nodeid: str, name: str, unverified: bool = False):
    """(group, file rank): 0 the kernel / shell tests of the last green suite, 1 its multi-rank tests, 2 verified-kernel tests
    written since, 3 multi-rank tests written since, 4 / 5 tests that reach device-unverified code (single process / multi-rank)."""
    fname = os.path.splitext(os.path.basename(nodeid.split("::")[0]))[0]
    base = name.split("[")[0]
    new = base in _NEW_SINCE_LAST_GREEN_SUITE or base.startswith("test_r5_") or base.startswith("test_r6_")
    multi = fname in _LAST_FILES
    rank = _LAST_FILES.index(fname) if multi else (_FILE_ORDER.index(fname) if fname in _FILE_ORDER else len(_FILE_ORDER))
    if unverified:
        return (5 if multi else 4, rank)
    return ((3 if new else 1) if multi else (2 if new else 0), rank)


def _hardware_verified_first(items) -> None:
    # stable: the order inside a group is the files' own
    items.sort(key=lambda it: _order_key(it.nodeid, it.name, it.get_closest_marker("device_unverified") is not None))


def never_run_kernels() -> set:
    """The kernels tools/kernel_table.py (the table of DESIGN.md section 5) lists as not run on an MI355X at HEAD."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_table", os.path.join(ROOT, "tools", "kernel_table.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return {k for k, row in mod.ROWS.items() if row[3].lstrip().startswith("**no**")}


def _gpu_twin_marked(item):
    """Has the GPU test this CPU-model test stands in for got the device_unverified marker?  None: no GPU twin (a test of
    the model itself, a CPU-only test) -- nothing to guard."""
    if "test_gpu_test_body_on_the_cpu_model[" in item.nodeid:                # tests/test_hip_emu.py runs tests/test_ops_gpu.py's bodies
        import tests.test_ops_gpu as G
        fn = getattr(G, item.callspec.params["name"], None)
        marks = getattr(fn, "pytestmark", []) if fn is not None else []
        return any(m.name == "device_unverified" for m in marks)
    if "[emu" in item.nodeid and hasattr(item, "callspec") and item.callspec.params.get("backend") == "emu":
        return item.get_closest_marker("device_unverified") is not None       # the same function, backend "hip", on the device
    return None


@pytest.fixture(autouse=True)
def _kernel_reach_guard(request):
    """See the ordering comment above: a CPU-model run of a GPU test that launches a never-run kernel without the
    device_unverified marker fails here."""
    import sys
    emu_ops = sys.modules.get("tests.hip_emu.emu_ops")
    marked = _gpu_twin_marked(request.node) if "[" in request.node.nodeid else None
    if marked is None:
        yield
        return
    before = emu_ops.launched_kernels() if emu_ops is not None else {}
    yield
    emu_ops = sys.modules.get("tests.hip_emu.emu_ops")
    if emu_ops is None:
        return
    after = emu_ops.launched_kernels()
    reached = {k for k, n in after.items() if n > before.get(k, 0)} & never_run_kernels()
    if reached and not marked:
        pytest.fail(f"{request.node.nodeid}: launched {sorted(reached)} -- kernels tools/kernel_table.py lists as never run on an "
                    "MI355X -- but its GPU twin does not carry @pytest.mark.device_unverified, so `pytest -x -m gpu` would "
                    "reach it before the verified kernels' tests (tests/conftest.py)")


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
