"""TEST INFRASTRUCTURE: one rank of `bench.py --gpus N` on the CPU (tests/test_bench_dryrun.py::test_two_ranks_print_one_line).

The same stand-ins as that file's `dry` fixture -- the product's HipOps over the STUB library (every C-ABI entry point returns at
once), torch.cuda's events / synchronisation replaced by host clocks, `torch.device("cuda", i)` = the CPU, sizes scaled down --
set with plain assignments (this is a process of its own), then bench.main() with the arguments given.  RANK / WORLD_SIZE /
MASTER_* and BDE_BENCH_BACKEND=gloo come from the caller, exactly as torch.distributed.run would set them.  Without them this is
`python bench.py` at N = 1: the GPU-free parent, whose headline / extras children are started through this same file
(test_the_whole_process_tree_of_the_default_run)."""
import contextlib
import importlib.util
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Event:
    def __init__(self, enable_timing=False):
        self.t = 0.0

    def record(self, stream=None):
        self.t = time.perf_counter()

    def synchronize(self):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return max((other.t - self.t) * 1e3, 1e-3)


class _Stream:
    cuda_stream = 0


class _TorchOnCpu:
    device = staticmethod(lambda *a, **k: torch.device("cpu"))
    empty_like = staticmethod(torch.zeros_like)

    def __getattr__(self, name):
        return getattr(torch, name)


def main():
    spec = importlib.util.spec_from_file_location("shell_host_cpu", os.path.join(ROOT, "tools", "shell_host_cpu.py"))
    host = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(host)
    import bench
    import beyond_deep_ensembles_amd.bbb_layers as L
    from beyond_deep_ensembles_amd import ops as ops_mod
    lib = host.stub_library()

    def init(self):
        self.lib = lib
        self.name = "hip_stub"
        self.load_code_objects = lambda device: None
    ops_mod.HipOps.__init__ = init
    ops_mod._ptr = lambda t, name="tensor": None if t is None else t.data_ptr()
    ops_mod._ptr64 = lambda t, name: t.data_ptr()
    ops_mod._stream = lambda: None
    L._native_nodes = lambda ops: None
    torch.cuda.Event = _Event
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda: None
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.device = lambda dev: contextlib.nullcontext()
    torch.cuda.set_device = lambda *a, **k: None
    bench.StreamProbes.__init__ = lambda self: setattr(self, "lib", None)
    bench.time_loop = lambda fn, iters, warm=3: (fn(), 1e-3)[1]
    bench.torch = _TorchOnCpu()
    # the flat-weight sizes scaled down (the code paths do not depend on them), also where they are default arguments
    small = {bench.D_RESNET50: 200_004, bench.D_DENSENET: 120_002}
    for fn in list(vars(bench).values()):
        if callable(fn) and getattr(fn, "__defaults__", None) and getattr(fn, "__module__", None) == "bench":
            fn.__defaults__ = tuple(small.get(v, v) if isinstance(v, int) else v for v in fn.__defaults__)
    bench.D_RESNET50, bench.D_DENSENET = 200_004, 120_002
    shapes = bench.resnet50_shapes
    bench.resnet50_shapes = lambda *a, **k: [tuple(v // 8 if v >= 64 else v for v in sh) for sh in shapes(*a, **k)]
    bench.__file__ = os.path.abspath(__file__)       # the parent's children re-enter through THIS file (same stand-ins)
    bench.main()


if __name__ == "__main__":
    main()
