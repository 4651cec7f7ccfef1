"""The one-line JSON contract of bench.py, checked on the bench lines kept under profiles/ (written on the MI355X
by the same bench.py): every key the driver and the judge read is there, with the units and bounds they expect."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lines():
    out = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_*bench*.json")) +
                       glob.glob(os.path.join(ROOT, "profiles", "r03_*bench*.json")) +
                       glob.glob(os.path.join(ROOT, "profiles", "r04_*bench*.json"))):
        with open(path) as fh:
            text = fh.read().strip().splitlines()[-1]
        out.append((os.path.basename(path), json.loads(text)))
    return out


def test_bench_lines_follow_the_contract():
    lines = _lines()
    assert lines, "no round-2 bench lines under profiles/"
    for name, d in lines:
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config"):
            assert key in d, (name, key)
        assert d["metric"] == "svgd_steps_per_s" and d["unit"] == "steps/s" and d["higher_is_better"] is True
        assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
        assert "workload" in d["config"] and "model" not in d["config"]
        assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 0.02 * d["value"]
        if d["n_gpus"] == 1:
            r = d["roofline"]
            for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
                assert key in r, (name, key)
            assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
            assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
            assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 0.01 * r["achieved"]
            if r["traffic"] is not None:                       # HBM traffic ~ algorithmic bytes: no wasted re-reads
                assert 0.99 <= r["traffic"] / r["algorithmic_bytes_per_launch"] <= 1.05
            # the dominant kernel cannot take longer than the step it is part of
            assert r["avg_launch_ms"] <= d["ms_per_step"]
            if "timing" in d:
                t = d["timing"]
                assert t["ms_per_step_min"] <= d["ms_per_step"] <= t["ms_per_step_max"]
                assert len(t["ms_per_step_blocks"]) == t["blocks"]
            if "cpu_baseline" in d:
                c = d["cpu_baseline"]
                for key in ("value", "unit", "cores", "kind", "sample"):
                    assert key in c, (name, key)
                assert c["kind"] in ("port", "reference") and c["cores"] >= 1
        else:
            assert d["config"]["exchange"] in ("alltoall", "pipelined", "allgather")
            if "headline" in d["exchange"]:
                # round 3: ONE invocation times every exchange mode; the headline is north_star's all-gather
                ex = d["exchange"]
                assert ex["headline"] == d["config"]["exchange"] == "allgather"
                assert {"allgather", "pipelined", "alltoall"} <= set(ex)
                for mode in ("allgather", "pipelined", "alltoall"):
                    assert {"exchange_ms", "update_ms", "step_ms", "overlap", "bytes_sent_per_rank"} <= set(ex[mode]), mode
                assert ex["best"] == min(("allgather", "pipelined", "alltoall"), key=lambda m: ex[m]["step_ms"])
                assert abs(ex["allgather"]["step_ms"] - d["ms_per_step"]) <= 1e-3 * d["ms_per_step"]
            else:
                assert {"exchange_ms", "update_ms", "step_ms", "overlap"} <= set(d["exchange"])


def test_bench_and_tools_call_hipops_with_existing_signatures():
    """bench.py, __graft_entry__.py and tools/*.py only run on the GPU box; every `ops.<method>(...)` call in them must at least
    name an existing HipOps method and bind to its signature (an API change in ops.py that forgets a caller shows up here, not
    as a missing bench entry at the end of a round)."""
    import ast
    import glob
    import inspect
    from beyond_deep_ensembles_amd.ops import HipOps
    checked, problems = 0, []
    for path in [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))):
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name)
                    and node.func.value.id == "ops"):
                continue
            fn = getattr(HipOps, node.func.attr, None)
            where = f"{os.path.relpath(path, ROOT)}:{node.lineno} ops.{node.func.attr}"
            if fn is None:
                problems.append(where + ": no such method")
                continue
            if isinstance(fn, property) or any(isinstance(a, ast.Starred) for a in node.args) or any(k.arg is None for k in node.keywords):
                continue
            try:
                inspect.signature(fn).bind(None, *[None] * len(node.args), **{k.arg: None for k in node.keywords})
                checked += 1
            except TypeError as e:
                problems.append(f"{where}: {e}")
    assert not problems, problems
    assert checked > 60
