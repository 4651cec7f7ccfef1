"""bench.py is 1200 lines that only run on the GPU box.  This runs its measurement functions end to end on the CPU with
the product's HipOps over a STUB library (every C-ABI entry point returns at once, tools/shell_host_cpu.py) and
torch.cuda's events / synchronisation replaced by host clocks: no kernel runs and no number means anything, but every
line of Python executes -- tensor shapes, argument lists, the optimizer shells' calls, the result dictionaries -- at the
real sizes.  An exception here is an `extras failed` / missing bench entry on the device."""
import importlib.util
import os
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


@pytest.fixture
def dry(monkeypatch):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
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
    monkeypatch.setattr(ops_mod.HipOps, "__init__", init)
    monkeypatch.setattr(ops_mod, "_ptr", lambda t, name="tensor": None if t is None else t.data_ptr())
    monkeypatch.setattr(ops_mod, "_ptr64", lambda t, name: t.data_ptr())
    monkeypatch.setattr(ops_mod, "_stream", lambda: None)
    monkeypatch.setattr(L, "_native_nodes", lambda ops: None)                # the C++ nodes bind the device library
    monkeypatch.setattr(torch.cuda, "Event", _Event)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, "device", lambda dev: __import__("contextlib").nullcontext())
    monkeypatch.setattr(bench.StreamProbes, "__init__", lambda self: setattr(self, "lib", None))   # the probe library is device code
    # every timed callable once (the CPU really executes the ATen reference sequences bench.py times beside the kernels),
    # and the flat-weight sizes scaled down 60x: the code paths do not depend on them
    monkeypatch.setattr(bench, "time_loop", lambda fn, iters, warm=3: (fn(), 1e-3)[1])
    small = {bench.D_RESNET50: 400_004, bench.D_DENSENET: 120_002}
    for fn in vars(bench).values():                                          # ... also where they are default arguments
        if callable(fn) and getattr(fn, "__defaults__", None) and getattr(fn, "__module__", None) == "bench":
            monkeypatch.setattr(fn, "__defaults__", tuple(small.get(v, v) if isinstance(v, int) else v for v in fn.__defaults__))
    monkeypatch.setattr(bench, "D_RESNET50", 400_004)
    monkeypatch.setattr(bench, "D_DENSENET", 120_002)
    shapes = bench.resnet50_shapes                                           # the 161 tensors, every channel count / 8
    monkeypatch.setattr(bench, "resnet50_shapes", lambda *a, **k: [tuple(v // 8 if v >= 64 else v for v in sh) for sh in shapes(*a, **k)])
    return bench, ops_mod.HipOps(), torch.device("cpu")


def test_extras_run_through(dry):
    bench, ops, dev = dry
    out = bench.extras(ops, dev, quick=True)
    for name in ("svgd_gram_M8_resnet50", "svgd_combine_M8_resnet50", "svgd_full_step_fused_sgd_M8_resnet20_2_launches",
                 "swag_update_resnet50", "swag_sample_K20_resnet50", "swag_sample_batched_K20_S30_resnet50",
                 "bbb_draw_fwd_resnet50", "bbb_kl_fwd_bwd_resnet50"):
        assert name in out and out[name]["ms"] > 0, name


def test_config_extras_and_shell_steps_run_through(dry):
    bench, ops, dev = dry
    cfg = bench.config_extras(dev)
    assert any(k.startswith("bbb_conv2d_fwd_bwd") for k in cfg), sorted(cfg)
    assert any(k.startswith("bbb_linear") for k in cfg), sorted(cfg)
    small = bench.shell_step_real_grads_ms(dev, n_tensors=12, d=120_000, steps=2)
    assert small["step_ms"] > 0
    ref = bench.shell_step_real_grads_ms(dev, ctor="reference", steps=1, particles=2)
    assert ref["step_ms"] > 0 and ref["tensors"] == 161
    assert bench.other_shell_steps_ms(dev, steps=1)


@pytest.mark.parametrize("launched", [False, True], ids=["python", "torchrun_one_rank"])
def test_main_prints_the_contract_line(dry, monkeypatch, capsys, launched):
    """main() itself at N = 1 (headline loop, roofline block, SWAG block, extras, shell steps) -> ONE JSON line with the keys
    the driver and the judge read (tests/test_bench_contract.py checks the same keys on the lines recorded on the device)."""
    import ctypes
    import json
    bench, ops, dev = dry

    class _TorchOnCpu:
        """bench.py's view of torch: `torch.device("cuda", i)` is the CPU, fresh buffers are zeroed (its finiteness check
        reads what the stub kernels never wrote)."""
        device = staticmethod(lambda *a, **k: torch.device("cpu"))
        empty_like = staticmethod(torch.zeros_like)

        def __getattr__(self, name):
            return getattr(torch, name)
    monkeypatch.setattr(bench, "torch", _TorchOnCpu())
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: None)
    real_cdll = ctypes.CDLL

    def cdll(path, *a, **k):
        if "bench_probe" in str(path):                                        # device code: a stand-in that "launches" nothing
            class _Fn:
                restype = argtypes = None

                def __call__(self, *args):
                    return 0
            return type("Probe", (), {"bde_bench_probe_r16w8": _Fn(), "bde_bench_probe": _Fn()})()
        return real_cdll(path, *a, **k)
    monkeypatch.setattr(ctypes, "CDLL", cdll)
    for key in ("RANK", "MASTER_ADDR", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(key, raising=False)
    if launched:          # as `torchrun --nproc-per-node 1 bench.py --gpus 1`: a process group of one rank (gloo here), extra.rccl_one_rank
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"),
                         ("MASTER_PORT", str(port)), ("BDE_BENCH_BACKEND", "gloo")):
            monkeypatch.setenv(key, val)
    # --extras-in-process: the stub library lives in THIS process (the default runs `extra` in a child process, below)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "2", "--warmup", "1", "--blocks", "3", "--no-cpu-baseline",
                                      "--extras-in-process", "--no-live-traffic"])
    bench.main()
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "swag", "extra"):
        assert key in d, key
    assert d["metric"] == "svgd_steps_per_s" and d["n_gpus"] == 1 and d["steps"] == 2 and "workload" in d["config"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    extra = d["extra"]
    assert "error" not in extra, extra.get("error")
    for key in ("svgd_combine_M8_resnet50", "swag_sample_batched_K20_S30_resnet50", "svgd_shell_step_ms",
                "svgd_shell_step_real_grads_ms", "svgd_reference_constructor_step", "other_shell_steps_ms", "other_baseline_configs"):
        assert key in extra, (key, sorted(extra))
    if launched:
        one = extra["rccl_one_rank"]
        for kind in ("allgather", "pipelined", "alltoall"):
            assert "error" not in one[kind] and one[kind]["step_ms"] > 0, (kind, one[kind])


def test_a_dying_extras_child_does_not_cost_the_line(dry, monkeypatch, tmp_path):
    """By default `extra` is measured by a child process (bench.extras_in_child): several extras launch kernels that have
    never run on an MI355X, and a GPU fault there must not take the headline line with it.  The sections a child finished
    before it died are kept, the death is reported under `error`; a child that hangs is killed (its exact PID) at the limit."""
    import argparse
    bench, ops, dev = dry
    args = argparse.Namespace(no_config_extras=True)
    script = tmp_path / "child.py"

    def run(body, limit=60):
        script.write_text("import json, os, sys, time\nout = sys.argv[sys.argv.index('--extras-child') + 1]\n" + body)
        monkeypatch.setattr(bench, "__file__", str(script))
        return bench.extras_in_child(args, 0, limit_s=limit)
    whole = run("json.dump({'svgd_combine_M8_resnet50': {'ms': 0.4}}, open(out, 'w'))\n")
    assert whole == {"svgd_combine_M8_resnet50": {"ms": 0.4}}
    died = run("json.dump({'a': 1}, open(out, 'w'))\nos.abort()\n")            # as a GPU fault ends a process: SIGABRT
    assert died["a"] == 1 and "exited with code" in died["error"]
    nothing = run("os._exit(3)\n")
    assert "exited with code 3" in nothing["error"]
    hung = run("json.dump({'b': 2}, open(out, 'w'))\ntime.sleep(600)\n", limit=2)
    assert "killed" in hung["error"] and hung["b"] == 2             # the sections a hung child had written are kept too


def test_the_real_extras_child_reports_sections_through_its_file(dry, monkeypatch, tmp_path):
    """single_gpu_extras(sink=...) -- what the child process runs -- rewrites the file after every section: the first block
    of kernels, each shell step, and LAST the sections that launch never-verified kernels."""
    import argparse
    import json
    bench, ops, dev = dry
    sink = tmp_path / "extra.json"
    seen = []
    real_shell = bench.shell_step_ms

    def shell(*a, **k):
        seen.append(sorted(json.load(open(sink))))                   # what had reached the file when the 2nd section began
        return real_shell(*a, **k)
    monkeypatch.setattr(bench, "shell_step_ms", shell)
    monkeypatch.setattr(bench, "shell_step_real_grads_ms", lambda *a, **k: {"step_ms": 1.0})
    monkeypatch.setattr(bench, "other_shell_steps_ms", lambda *a, **k: {"tensors": 1})
    ex = bench.single_gpu_extras(ops, dev, argparse.Namespace(no_config_extras=True), sink=str(sink))
    assert seen and "svgd_combine_M8_resnet50" in seen[0] and "svgd_full_step_fused_sgd_M8_resnet20_2_launches" in seen[0]
    assert json.load(open(sink)) == json.loads(json.dumps(ex))
    order = list(ex)
    assert order.index("svgd_combine_M8_resnet50") < order.index("svgd_step_M8_resnet20")       # never-verified kernels last


def test_live_traffic_parsing_and_fallback(dry, monkeypatch, tmp_path):
    """bench.live_traffic: roofline.traffic from two `rocprofv3 --pmc` child passes of THIS run.  (1) The counter arithmetic on
    the round-1 PMC files committed under profiles/ (FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled on gfx950) reproduces
    the per-launch bytes of the combine kernel recorded then, 1.000x its 12 M D algorithmic bytes.  (2) Without a working
    profiler / GPU the function reports why and the recorded value stays."""
    bench, ops, dev = dry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fetch, nf = bench.parse_pmc_counter([os.path.join(root, "profiles", "r01_b_pmc_FETCH_SIZE.csv")], "FETCH_SIZE", "svgd_combine_kernel")
    write, nw = bench.parse_pmc_counter([os.path.join(root, "profiles", "r01_b_pmc_WRITE_SIZE.csv")], "WRITE_SIZE", "svgd_combine_kernel")
    assert nf >= 3 and nw >= 3
    nbytes = bench.pmc_traffic_bytes(fetch, write)
    assert abs(nbytes / (12 * 8 * 23_880_950) - 1.0) < 2e-3, nbytes
    assert bench.parse_pmc_counter([os.path.join(root, "profiles", "r01_b_pmc_FETCH_SIZE.csv")], "FETCH_SIZE", "no_such_kernel") == (None, 0)
    # a profiler that fails / is absent / a profiled parent: (None, reason)
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/sh\nexit 7\n")
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ.get("PATH", ""))
    got, why = bench.live_traffic(1000, 0, limit_s=20)
    assert got is None and "exited with code 7" in why
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.live_traffic(1000, 0)[1] == "this process is itself being profiled"

