"""mp.spawn for tests that run SEVERAL ranks on ONE GPU (a harness-only situation: the product runs one process per GPU).

Several processes sharing a device can, rarely, lose a process to the HIP runtime's queue abort
``HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION`` (SIGABRT) -- seen since round 2 with 8 ranks on one device, at a rate of about
one run in 25-45 even with every code object of libbde_hip.so loaded up front (bde_init()).  DESIGN.md section 6 has the
evidence and what round 4's fault hunt (tools/fault_hunt.sh) found.  The harness re-runs the ranks ONCE when EXACTLY that
happens -- a rank died with SIGABRT *and* the ranks' stderr carries the runtime's HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION
queue-abort line -- records the event (``RETRIES`` here, ``gpurun_out/one_device_retries.log`` when that directory
exists, and pytest's terminal summary through tests/conftest.py), and a SECOND such event in one pytest session fails
the test.  Any other SIGABRT (a collective watchdog, a C++ assertion, a HIP launch failure), a Python exception in a
rank, a wrong result: the test fails as usual.
"""
import os
import sys
import tempfile
import time
import warnings

import torch.multiprocessing as mp
from torch.multiprocessing.spawn import ProcessExitedException

FAULT_TEXT = "HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION"
MAX_RETRIES_PER_SESSION = 1
RETRIES = []            # one entry per retried spawn of this pytest session: (time, nprocs, the runtime's abort line)


def _spawn_capturing_stderr(fn, args, nprocs):
    """mp.spawn with file descriptor 2 of this process (inherited by the ranks) redirected into a file for the duration;
    returns (exception or None, what the ranks and this process wrote to stderr meanwhile).  The text is passed on to the
    real stderr afterwards, so nothing is hidden from the test log."""
    sys.stderr.flush()
    saved = os.dup(2)
    with tempfile.TemporaryFile(mode="w+b") as cap:
        os.dup2(cap.fileno(), 2)
        exc = None
        try:
            mp.spawn(fn, args=args, nprocs=nprocs, join=True)
        except BaseException as e:          # noqa: BLE001 -- re-raised by the caller
            exc = e
        finally:
            sys.stderr.flush()
            os.dup2(saved, 2)
            os.close(saved)
        cap.seek(0)
        text = cap.read().decode("utf-8", "replace")
    if text:
        sys.stderr.write(text)
        sys.stderr.flush()
    return exc, text


def spawn_ranks(fn, make_args, nprocs):
    """``make_args()`` -> the args tuple (called per attempt: a rendezvous port must be fresh)."""
    exc, text = _spawn_capturing_stderr(fn, make_args(), nprocs)
    if exc is None:
        return
    is_fault = isinstance(exc, ProcessExitedException) and getattr(exc, "signal_name", None) == "SIGABRT" \
        and FAULT_TEXT in text
    if not is_fault:
        raise exc
    line = next((ln for ln in text.splitlines() if FAULT_TEXT in ln), "")[:300]
    RETRIES.append((time.strftime("%Y-%m-%d %H:%M:%S"), nprocs, line))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "one_device_retries.log"), "a") as f:
            f.write(f"{RETRIES[-1][0]} nprocs={nprocs} test={os.environ.get('PYTEST_CURRENT_TEST', '?')} :: {line}\n")
    if len(RETRIES) > MAX_RETRIES_PER_SESSION:
        raise RuntimeError(f"{len(RETRIES)} HIP queue aborts ({FAULT_TEXT}) in one test session: more than the one the "
                           "several-ranks-on-one-device harness tolerates") from exc
    warnings.warn(f"a rank died with SIGABRT and the runtime reported {FAULT_TEXT} ({line}): the HIP runtime's queue abort "
                  "with several processes on one device; running the ranks once more")
    exc, _ = _spawn_capturing_stderr(fn, make_args(), nprocs)
    if exc is not None:
        raise exc
