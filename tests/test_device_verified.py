"""beyond_deep_ensembles_amd/device_verified.py: the gate between code that has been green on an MI355X and code that has not
(VERDICT r5 #3).  CPU tests of the gate itself; what it switches is covered by the shells' tests."""
import json
import os

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import beyond_deep_ensembles_amd as bde
from beyond_deep_ensembles_amd import device_verified as V
from tests.oracle_ops import OracleOps


def test_the_committed_table_is_consistent_with_the_sources():
    """Every record of the committed table belongs to a known family; a family counts as verified exactly when its record's hash
    is the hash of the sources in the tree (today: no record at all -- no GPU has run this tree)."""
    table = json.load(open(os.path.join(os.path.dirname(V.__file__), "device_verified.json")))
    assert set(table["families"]) <= set(V.FAMILIES)
    for fam in V.FAMILIES:
        assert len(V.source_hash(fam)) == 64                         # all sources of the family are in the package
        rec = table["families"].get(fam)
        assert V.enabled(fam, table) == (rec is not None and rec.get("sha256") == V.source_hash(fam))
    with pytest.raises(KeyError):
        V.enabled("no_such_family")


def test_a_record_opens_the_gate_only_for_the_sources_it_was_written_for(monkeypatch, tmp_path):
    monkeypatch.delenv("BDE_UNVERIFIED", raising=False)
    monkeypatch.setattr(V, "_PATH", str(tmp_path / "device_verified.json"))
    monkeypatch.setattr(V, "_table", None)
    assert not V.enabled("svgd_small") and V.status()["svgd_small"] == "unverified (no record)"
    rec = V.record("svgd_small", device="AMD Instinct MI355X", tests_passed=12, log="profiles/r06_verify_svgd_small.log")
    assert rec["sha256"] == V.source_hash("svgd_small")
    assert V.enabled("svgd_small") and not V.enabled("mean_scalars") and V.status()["svgd_small"] == "verified"
    # the sources change (a kernel edit after the device run): the record no longer counts
    monkeypatch.setitem(V._hashes, "svgd_small", "0" * 64)
    assert not V.enabled("svgd_small") and V.status()["svgd_small"] == "unverified (sources changed since the record)"
    # the A/B override names families explicitly, or all of them
    monkeypatch.setenv("BDE_UNVERIFIED", "mean_scalars")
    assert V.enabled("mean_scalars") and not V.enabled("svgd_small")
    monkeypatch.setenv("BDE_UNVERIFIED", "all")
    assert all(V.enabled(f) for f in V.FAMILIES)


def test_the_gate_decides_what_a_default_constructed_optimizer_runs(monkeypatch, tmp_path):
    """SVGDOptimizer(single_launch=None, host_fast_paths=None) on a small model: the streaming kernels + torch's loss adds + the
    begin / end particle loop without records; the small-model kernel + one-launch loss mean + the fast loop with them;
    identical particles either way (the checker backend computes both forms with the oracle)."""
    monkeypatch.delenv("BDE_UNVERIFIED", raising=False)
    monkeypatch.setattr(V, "_PATH", str(tmp_path / "device_verified.json"))
    monkeypatch.setattr(V, "_table", None)
    torch.manual_seed(0)
    x, y = torch.randn(16, 13), torch.randn(16, 1)

    def run():
        torch.manual_seed(1)
        model = nn.Sequential(nn.Linear(13, 20), nn.Tanh(), nn.Linear(20, 1))
        ops = OracleOps()
        calls = {"small": 0, "stream": 0, "mean": 0}
        for name, key in (("svgd_step_small_sgd", "small"), ("svgd_fused_sgd_seg", "stream"), ("mean_scalars", "mean")):
            real = getattr(ops, name)
            setattr(ops, name, lambda *a, _r=real, _k=key, **k: (calls.__setitem__(_k, calls[_k] + 1), _r(*a, **k))[1])
        base = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
        opt = bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model), base, particle_count=4, dataset_size=16,
                                _ops=ops)
        for _ in range(3):
            opt.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward())
        return opt.particles.clone(), calls, opt
    p0, calls0, opt0 = run()
    assert calls0 == {"small": 0, "stream": 3, "mean": 0} and not opt0._gate("fast_loop")
    for fam in V.FAMILIES:
        V.record(fam, device="AMD Instinct MI355X", tests_passed=1)
    p1, calls1, opt1 = run()
    assert calls1["small"] == 3 and calls1["stream"] == 0 and calls1["mean"] == 3 and opt1._gate("fast_loop")
    torch.testing.assert_close(p1, p0, rtol=1e-5, atol=1e-6)
