"""The several-ranks-on-one-device harness itself (tests/spawn_one_device.py): it retries ONLY the HIP runtime's
HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION queue abort, once, and counts it; any other abort fails."""
import os
import sys

import pytest
from torch.multiprocessing.spawn import ProcessExitedException

from tests import spawn_one_device as sod


def _worker(rank, marker, say_fault):
    if rank == 1 and not os.path.exists(marker):
        open(marker, "w").close()
        if say_fault:
            sys.stderr.write(":0:rocdevice.cpp :3676: Callback: Queue 0x1 aborting with error : "
                             "HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION: The agent attempted to execute an illegal shader "
                             "instruction. code: 0x2a\n")
            sys.stderr.flush()
        os.abort()


def test_retries_exactly_the_runtime_queue_abort_once(tmp_path, monkeypatch):
    monkeypatch.setattr(sod, "RETRIES", [])
    with pytest.warns(UserWarning, match="HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION"):
        sod.spawn_ranks(_worker, lambda: (str(tmp_path / "a"), True), 2)
    assert len(sod.RETRIES) == 1 and "rocdevice.cpp" in sod.RETRIES[0][2]
    # a second event in the same session is not tolerated
    with pytest.raises(RuntimeError, match="more than the one"):
        sod.spawn_ranks(_worker, lambda: (str(tmp_path / "b"), True), 2)


def test_any_other_abort_fails(tmp_path, monkeypatch):
    monkeypatch.setattr(sod, "RETRIES", [])
    with pytest.raises(ProcessExitedException):
        sod.spawn_ranks(_worker, lambda: (str(tmp_path / "c"), False), 2)
    assert sod.RETRIES == []


def test_shells_without_the_native_host_helper():
    """lib/_bde_host.so only moves the shells' per-tensor loops and the layers' autograd nodes into C++; without it
    (`__graft_entry__.build()` warns and carries on if it cannot be built) the same loops run in Python (algo.py:
    repoint / collect_grads / adopt_grads; bbb_layers.py: the Python autograd Functions).  The trajectory, schedule,
    checkpoint and GradScaler tests of the shells once more with BDE_NO_HOST_HELPER=1, in a child process (the helper is
    loaded once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BDE_NO_HOST_HELPER="1")
    sel = ("oracle and (svgd_trajectory or swag_schedule or swag_statistics or ivon_trajectory or bbb_trajectory or state_dict or "
           "grad_scaler or cnn_training or many_particles or streaming or conv_layers or fused_path)")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_shells.py"), "-q", "-x", "-k", sel,
                        "-p", "no:cacheprovider"], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = p.stdout.decode()[-1500:]
    assert p.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
