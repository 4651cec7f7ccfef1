"""Build-container tool (needs /root/reference; ~3 min): BASELINE configs[4] at its real parameter count -- five SwagOptimizer members of
D = 6,955,906 (Camelyon DenseNet-121), K = 20, 22 updates each, 10 predictions through DeepEnsemble.predict (ensemble.py:28-44) -- the
imported reference next to ours over the kernel sources on the CPU model (tests/hip_emu).  Arithmetic only, not a device run."""
import os, sys, time, math, torch, torch.nn as nn
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "/root/reference")
import src.algos.swag as rswag, src.algos.ensemble as rens
import beyond_deep_ensembles_amd as bde
from tests.hip_emu import emu_ops
torch.set_num_threads(os.cpu_count())
D, K, UPD, MEMBERS, SAMPLES = 6_955_906, 20, 22, 5, 10
class Vec(nn.Module):
    def __init__(self, v):
        super().__init__(); self.v = nn.Parameter(v.clone())
def build(side, ops=None):
    out = []
    for mem in range(MEMBERS):
        g = torch.Generator().manual_seed(100 + mem)
        model = Vec(torch.randn(D, generator=g) * 0.05)
        base = torch.optim.SGD(model.parameters(), lr=1.0)
        kw = dict(update_interval=1, start_epoch=0, deviation_samples=K)
        opt = rswag.SwagOptimizer(model.parameters(), base, **kw) if side == "ref" else bde.SwagOptimizer(model.parameters(), base, _ops=ops, **kw)
        for t in range(UPD):
            w = torch.randn(D, generator=g) * 1e-3
            opt.step(lambda: -(model.v * w).sum(), lambda l: l.backward())
        out.append((model, opt))
    return out
probe = torch.randn(4, D, generator=torch.Generator().manual_seed(5)) / D ** 0.5
closure = lambda model: torch.log_softmax(probe @ model.v.detach() * 50, dim=0)      # a 4-class "prediction" that depends on every weight
t0 = time.time(); theirs = rens.DeepEnsemble(build("ref")); torch.manual_seed(9); want = theirs.predict(closure, SAMPLES); t1 = time.time()
print(f"reference: {t1-t0:.0f} s", flush=True)
with emu_ops.emulated(emu_ops.ALL) as ops:
    ours = bde.DeepEnsemble(build("ours", ops)); torch.manual_seed(9); got = ours.predict(closure, SAMPLES)
    same = all(torch.equal(a[1].state["__mean"], b[1].mean_vector().cpu()) and torch.equal(a[1].state["__deviations"], b[1].deviations_dk().cpu()) for a, b in zip(theirs.models_and_optimizers, ours.models_and_optimizers))
print(f"MultiSWAG, {MEMBERS} members x D = {D:,} (Camelyon DenseNet-121 size), K = {K}, {UPD} updates each, {SAMPLES} predictions (reference {t1-t0:.0f} s, CPU model {time.time()-t1:.0f} s)")
print(f"  every member's mean and [D, K] deviations bit-exact: {same}")
print(f"  predictions [S, 4]: max |ours - reference| {float((got-want).abs().max()):.2e} (max |.| {float(want.abs().max()):.2e}); logsumexp(out, 0) - log S: {float(((torch.logsumexp(got,0)-math.log(SAMPLES))-(torch.logsumexp(want,0)-math.log(SAMPLES))).abs().max()):.2e}")
