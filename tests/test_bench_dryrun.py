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
    assert not {k: v for k, v in out.items() if isinstance(v, dict) and "error" in v}
    assert "svgd_full_step_fused_sgd_M8_resnet20_streaming" in out and "svgd_step_M8_resnet20_small_kernel" in out
    for name in ("svgd_gram_M8_resnet50", "svgd_combine_M8_resnet50", "svgd_full_step_fused_sgd_M8_resnet20_2_launches",
                 "swag_update_resnet50", "swag_sample_K20_resnet50", "swag_sample_batched_K20_S30_resnet50",
                 "bbb_draw_fwd_resnet50", "bbb_kl_fwd_bwd_resnet50"):
        assert name in out and out[name]["ms"] > 0, name


def test_config_extras_and_shell_steps_run_through(dry):
    bench, ops, dev = dry
    cfg = bench.config_extras(dev)
    failed = {k: v for k, v in cfg.items() if isinstance(v, dict) and ("error" in v or "error" in v.get("fused_kernels", {}))}
    assert not failed, failed                                                # (sections catch their own exceptions: look inside)
    for key in ("svgd_step_cifar_resnet20_shell_fused", "svgd_step_cifar_resnet20_shell_fused_small_kernel",
                "svgd_step_cifar_resnet20_shell_unfused_small_kernel", "svgd_step_cifar_resnet20_shell_fused_graph_replay"):
        assert cfg[key]["ms"] > 0, key
    from beyond_deep_ensembles_amd import device_verified             # the default: device-verified kernels only
    assert cfg["svgd_step_cifar_resnet20_shell_fused"]["small_model_kernel"] is device_verified.enabled("svgd_small")
    assert cfg["svgd_step_cifar_resnet20_shell_fused_small_kernel"]["small_model_kernel"] is True
    assert any(k.startswith("bbb_conv2d_fwd_bwd") for k in cfg), sorted(cfg)
    assert all("fused_kernels" in v for k, v in cfg.items() if k.startswith("bbb_conv2d_fwd_bwd"))
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



# ---- the line is unlosable ----------------------------------------------------------------------------
_FAKE_BENCH = '''
import json, os, sys, time
argv = sys.argv
def arg(name):
    return argv[argv.index(name) + 1]
mode = os.environ.get("FAKE_MODE", "ok")
if "--headline-child" in argv:
    assert os.environ.get("EXPECT_RANK") == os.environ.get("RANK"), "the headline child must inherit the rendezvous variables"
    res = {"metric": "svgd_steps_per_s", "value": 1234.5, "unit": "steps/s", "n_gpus": 1, "steps": int(arg("--steps")),
           "warmup": int(arg("--warmup")), "ms_per_step": 0.81, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic", "config": {"workload": "fake"},
           "roofline": {"bound": "hbm", "achieved": 5600.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.7, "traffic": 7, "traffic_source": "recorded"}}
    if mode == "headline_dies_at_once":
        os.abort()
    json.dump(res, open(arg("--headline-child"), "w"))
    if mode == "headline_dies_after_the_timed_region":
        os.abort()                                         # as a GPU fault in an optional section ends the process
    res["swag"] = {"samples_per_s": 2700.0}
    res["rccl_one_rank"] = {"allgather": {"step_ms": 1.0}}
    json.dump(res, open(arg("--headline-child"), "w"))
elif "--extras-child" in argv:
    assert "RANK" not in os.environ and "MASTER_ADDR" not in os.environ, "the extras child is no rank"
    json.dump({"svgd_combine_M8_resnet50": {"ms": 0.4}}, open(arg("--extras-child"), "w"))
    if mode == "extras_die":
        os.abort()
'''


def _orchestrate(bench, monkeypatch, tmp_path, capsys, mode="ok", argv=()):
    import json
    script = tmp_path / "fake_bench.py"
    script.write_text(_FAKE_BENCH)
    monkeypatch.setattr(bench, "__file__", str(script))
    monkeypatch.setenv("FAKE_MODE", mode)
    monkeypatch.setattr(bench, "cpu_baseline", lambda P, G, d: {"value": 0.28, "unit": "steps/s", "cores": 8, "kind": "port", "sample": "x"})
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "20", "--warmup", "5", "--no-live-traffic", *argv])
    with pytest.raises(SystemExit) as done:
        bench.main()
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    return done.value.code, json.loads(lines[0])


@pytest.mark.parametrize("launched", [False, True], ids=["python", "torchrun_one_rank"])
def test_the_default_run_is_a_gpu_free_parent_that_merges_its_children(dry, monkeypatch, tmp_path, capsys, launched):
    """`python bench.py` (N = 1): the parent creates the headline child, times the CPU baseline itself, creates the extras child
    and prints ONE line merged from what they wrote -- and it never touches the GPU itself (any torch.cuda call fails here)."""
    bench, ops, dev = dry
    for key in ("RANK", "MASTER_ADDR", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(key, raising=False)

    def touched(*a, **k):
        raise AssertionError("the parent process initialised the GPU")
    for name in ("set_device", "synchronize", "current_device", "is_available", "Event", "current_stream", "empty_cache"):
        monkeypatch.setattr(torch.cuda, name, touched)
    if launched:
        # `torchrun --nproc-per-node 1 bench.py --gpus 1`: the headline child IS the rank (it inherits the rendezvous variables
        # and joins the process group); the extras child is an ordinary process without them
        for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29999"),
                         ("EXPECT_RANK", "0")):
            monkeypatch.setenv(key, val)
    code, line = _orchestrate(bench, monkeypatch, tmp_path, capsys)
    assert code == 0 and line["value"] == 1234.5 and line["steps"] == 20 and line["warmup"] == 5
    assert line["roofline"]["frac"] == 0.7 and line["cpu_baseline"]["kind"] == "port" and line["swag"]["samples_per_s"] == 2700.0
    assert line["extra"]["svgd_combine_M8_resnet50"] == {"ms": 0.4}
    assert line["extra"]["rccl_one_rank"] == {"allgather": {"step_ms": 1.0}} and "rccl_one_rank" not in line


@pytest.mark.parametrize("mode", ["headline_dies_after_the_timed_region", "extras_die", "cpu_baseline_raises", "popen_refused",
                                  "popen_refused_for_the_extras", "headline_dies_at_once"])
def test_no_failing_part_costs_the_line(dry, monkeypatch, tmp_path, capsys, mode):
    """Every optional part may fail -- a GPU fault in a child behind the timed region, a dying extras child, an exception in the
    CPU baseline, a box that refuses to create processes (PermissionError from Popen) -- and the line is still printed with
    what was measured and the failure recorded beside it.  Only a missing HEADLINE makes the exit code non-zero, and even then
    a line (value null, `error`) is printed."""
    import subprocess
    bench, ops, dev = dry
    for key in ("RANK", "MASTER_ADDR", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(key, raising=False)
    real_popen = subprocess.Popen
    if mode.startswith("popen_refused"):
        def popen(cmd, *a, **k):
            if mode == "popen_refused" or "--extras-child" in cmd:
                raise PermissionError(1, "Operation not permitted")
            return real_popen(cmd, *a, **k)
        monkeypatch.setattr(subprocess, "Popen", popen)
    if mode == "cpu_baseline_raises":
        script_mode = "ok"
    else:
        script_mode = mode
    code, line = _orchestrate(bench, monkeypatch, tmp_path, capsys, mode=script_mode) if mode != "cpu_baseline_raises" else (None, None)
    if mode == "cpu_baseline_raises":
        def boom(P, G, d):
            raise MemoryError("cannot allocate the host copy")
        script = tmp_path / "fake_bench.py"
        script.write_text(_FAKE_BENCH)
        monkeypatch.setattr(bench, "__file__", str(script))
        monkeypatch.setenv("FAKE_MODE", "ok")
        monkeypatch.setattr(bench, "cpu_baseline", boom)
        monkeypatch.setattr(sys, "argv", ["bench.py", "--no-live-traffic"])
        with pytest.raises(SystemExit) as done:
            bench.main()
        import json
        line = json.loads([ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")][0])
        assert done.value.code == 0 and line["value"] == 1234.5 and "MemoryError" in line["cpu_baseline"]["error"]
        assert line["extra"]["svgd_combine_M8_resnet50"] == {"ms": 0.4}
    elif mode == "headline_dies_after_the_timed_region":
        assert code == 0 and line["value"] == 1234.5 and "exited with code" in line["headline_child_error"]
        assert "swag" not in line and line["extra"]["svgd_combine_M8_resnet50"] == {"ms": 0.4}
    elif mode == "extras_die":
        assert code == 0 and line["value"] == 1234.5 and "exited with code" in line["extra"]["error"]
        assert line["extra"]["svgd_combine_M8_resnet50"] == {"ms": 0.4}            # the sections it had written are kept
    elif mode == "popen_refused_for_the_extras":
        assert code == 0 and line["value"] == 1234.5 and "PermissionError" in line["extra"]["error"]
        assert line["extra"]["rccl_one_rank"] == {"allgather": {"step_ms": 1.0}}
    else:
        assert code == 1 and line["value"] is None and line["metric"] == "svgd_steps_per_s"
        assert ("PermissionError" if mode == "popen_refused" else "exited with code") in line["error"]


def test_sections_behind_the_timed_region_are_guarded_and_checkpointed(dry, monkeypatch, capsys):
    """bench.headline(sink=...) -- what the headline child runs: the result is handed to the sink as soon as the timed region
    and the roofline exist and again after every optional section; a section that raises (here: the probe library, the SWAG
    block, the reference-on-GPU baseline) becomes an `error` record in its place."""
    import json
    bench, ops, dev = dry

    class _TorchOnCpu:
        device = staticmethod(lambda *a, **k: torch.device("cpu"))
        empty_like = staticmethod(torch.zeros_like)

        def __getattr__(self, name):
            return getattr(torch, name)
    monkeypatch.setattr(bench, "torch", _TorchOnCpu())
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: None)
    for key in ("RANK", "MASTER_ADDR", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(key, raising=False)

    def boom(*a, **k):
        raise RuntimeError("HIP error: an illegal memory access was encountered")
    monkeypatch.setattr(bench, "stream_probe", boom)
    monkeypatch.setattr(bench, "torch_gpu_baseline", boom)
    monkeypatch.setattr(type(ops), "swag_sample_batched", boom)
    seen = []
    args = bench.parse_args(["--steps", "2", "--warmup", "1", "--blocks", "2", "--no-extras"])
    res = bench.headline(args, sink=lambda r: seen.append(json.loads(json.dumps(r))))
    assert len(seen) >= 3 and all(s["value"] == seen[0]["value"] > 0 for s in seen)
    assert "roofline" in seen[0] and "swag" not in seen[0]                       # the first hand-over precedes every optional section
    assert "illegal memory access" in res["roofline"]["probe_error"]
    assert "illegal memory access" in res["swag"]["error"] and "illegal memory access" in res["gpu_torch_baseline"]["error"]
    assert "cpu_baseline" not in res and "extra" not in res                    # the parent's parts
    assert seen[-1] == json.loads(json.dumps(res))


@pytest.mark.parametrize("world", [2, 8])
def test_two_ranks_print_one_line(tmp_path, world):
    """`torch.distributed.run --nproc-per-node N bench.py --gpus N` (N = 2, 8) as the driver's scaling run launches it, on the CPU: N
    processes (tests/bench_dry_rank.py: stub library, gloo), the product's multi-GPU update in all three exchange modes timed by
    each rank, barriers and the MAX reduction over ranks -- rank 0 prints the ONE line (n_gpus 2, scaling "strong", `exchange`
    with every mode), rank 1 prints nothing, both exit 0."""
    import json
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BDE_BENCH_BACKEND="gloo", BDE_BENCH_DEVICE="0", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "bench_dry_rank.py"), "--gpus", str(world), "--steps", "2",
                                       "--warmup", "1", "--blocks", "2", "--dim", "200004"], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=600) for p in procs]
    assert [p.returncode for p in procs] == [0] * world, (outs[0][1].decode()[-2000:], outs[-1][1].decode()[-2000:])
    lines = [ln for ln in outs[0][0].decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for o in outs[1:] for ln in o[0].decode().splitlines() if ln.startswith("{")]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["value"] > 0 and d["config"]["particles_per_rank"] == 8 // world
    assert set(d["exchange"]) >= {"allgather", "pipelined", "alltoall", "headline", "best"} and d["exchange"]["headline"] == "allgather"
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    for mode in ("allgather", "pipelined", "alltoall"):
        assert d["exchange"][mode]["step_ms"] > 0, mode
    # every rank's stand-in rate (1 ms per call) summed over the ranks: the aggregate is the whole job's
    assert d["swag"]["samples_per_s"] == 1000.0 * world and d["swag"]["samples_per_s_batched_S30"] == 30000.0 * world


def test_the_whole_process_tree_of_the_default_run(tmp_path):
    """`python bench.py` at N = 1 with REAL children: tests/bench_dry_rank.py is bench.main() over the CPU stand-ins, and the
    parent it becomes starts its headline and extras children through the same file -- parent -> `--headline-child` (timed
    region, roofline, SWAG, the reference's op sequence) -> CPU baseline in the parent -> `--extras-child` (every section) ->
    ONE merged line.  The probe library is device code: its launch fails here (-2), which must show up as
    roofline.probe_error beside an intact line -- the guard of the first optional section, exercised for real."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_dry_rank.py"), "--steps", "2", "--warmup", "1", "--blocks", "2",
                        "--no-live-traffic"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and "headline_child_error" not in d
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    if os.path.exists(os.path.join(ROOT, "bench_probe", "libbde_bench_probe.so")):
        assert "probe_error" in d["roofline"] and "frac_of_probe" not in d["roofline"]
    assert d["swag"]["samples_per_s"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["gpu_torch_baseline"]["svgd_steps_per_s"] > 0
    extra = d["extra"]
    assert "error" not in extra, extra.get("error")
    for key in ("svgd_combine_M8_resnet50", "swag_sample_batched_K20_S30_resnet50", "svgd_shell_step_ms", "svgd_shell_step_real_grads_ms",
                "svgd_reference_constructor_step", "other_shell_steps_ms", "other_baseline_configs"):
        assert key in extra, (key, sorted(extra))
    assert not [k for k, v in extra.items() if isinstance(v, dict) and "error" in v]
