"""mp.spawn for tests that run SEVERAL ranks on ONE GPU (a harness-only situation: the product runs one process per GPU).

Several processes sharing a device can, rarely, lose a process to the HIP runtime's queue abort
``HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION`` (SIGABRT) -- seen since round 2 with 8 ranks on one device, at a rate of about
one run in 25 even with every code object of libbde_hip.so loaded up front (bde_init(); HIP's own switch
HIP_ENABLE_DEFERRED_LOADING=0 makes this torch build segfault at start-up, so torch's own code objects cannot be preloaded).
DESIGN.md section 6 has the evidence.  It is not reproducible on demand, so the harness
re-runs the ranks ONCE when exactly that happens and says so; anything else -- a Python exception in a rank, a wrong
result, a second abort -- fails the test as usual.
"""
import warnings

import torch.multiprocessing as mp
from torch.multiprocessing.spawn import ProcessExitedException


def spawn_ranks(fn, make_args, nprocs):
    """``make_args()`` -> the args tuple (called per attempt: a rendezvous port must be fresh)."""
    for attempt in (0, 1):
        try:
            mp.spawn(fn, args=make_args(), nprocs=nprocs, join=True)
            return
        except ProcessExitedException as exc:
            if attempt == 0 and getattr(exc, "signal_name", None) == "SIGABRT":
                warnings.warn(f"a rank died with SIGABRT ({exc}): the HIP runtime's queue abort with several processes on "
                              "one device; running the ranks once more")
                continue
            raise
