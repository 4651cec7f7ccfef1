"""Shared body of the multi-rank checkpoint tests (tests/test_dist_cpu.py with the oracle checker on CPU,
tests/test_dist_gpu.py with HipOps on cuda:0): a state_dict WRITTEN BY THE REFERENCE SVGDOptimizer (four particles,
nesterov SGD with weight decay; tests/golden/ref_svgd_checkpoint4.pt, made by oracle/gen_golden.py) is loaded into a
2-rank optimizer in the given exchange mode, resumed for one step, saved with the optimizer's own state_dict() (a
collective with exchange="alltoall"), loaded into a FRESH optimizer and resumed for a second step.  Both steps must
land on the reference's own next steps (src/algos/svgd.py:65-105; checkpoints: src/algos/ensemble.py:17-26,
experiments/iwildcam/iwildcam.py:84-88,161)."""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_mlp():
    return nn.Sequential(nn.Linear(13, 50), nn.ReLU(), nn.Linear(50, 1))


def build(ops, dev, pg, kw, model_state):
    import beyond_deep_ensembles_amd as bde
    model = make_mlp().to(dev)
    model.load_state_dict(model_state)
    base = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4)
    opt = bde.SVGDOptimizer(model.parameters(), lambda: None, base, particle_count=4, dataset_size=64, l2_reg=0.01,
                            process_group=pg, fuse_base_optimizer=True, _ops=ops, **kw)
    return model, base, opt


def resume_worker(rank, ops, dev, pg, kw, out_dir):
    ck = torch.load(os.path.join(GOLD, "ref_svgd_checkpoint4.pt"), weights_only=False)
    x, y = ck["x"].to(dev), ck["y"].to(dev)
    model, base, opt = build(ops, dev, pg, kw, ck["model"])
    ref_base = ck["optimizer"]["state"]["__base_optimizer"]            # the reference's pickled torch.optim.SGD
    sd = ck["optimizer"]
    sd["state"]["__base_optimizer"] = base                             # the caller owns the base optimizer
    opt.load_state_dict(sd)
    loaded = opt.particles.detach().cpu().clone().numpy()
    # the shared momentum buffers of the reference's base optimizer (keyed on ITS parameters) carried over
    for p_new, p_old in zip(model.parameters(), ref_base.param_groups[0]["params"]):
        base.state[p_new]["momentum_buffer"] = ref_base.state[p_old]["momentum_buffer"].clone().to(dev)
    loss1 = float(opt.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward()))
    after1 = opt.particles.detach().cpu().clone().numpy()
    # our own checkpoint, in the reference's layout, through a file
    # (the way the reference's drivers save: torch.save(ensemble.state_dict()), iwildcam.py:161 -> ensemble.py:17-26)
    import beyond_deep_ensembles_amd as bde
    mine = bde.DeepEnsemble([(model, opt)]).state_dict()
    keys = sorted(k for k in mine["optimizers"][0]["state"][0] if str(k).startswith("particle_"))
    path = os.path.join(out_dir, f"ckpt_rank{rank}.pt")
    torch.save(mine, path)
    saved = torch.load(path, weights_only=False)
    model2, base2, opt2 = build(ops, dev, pg, kw, ck["model"])
    sd2 = saved["optimizers"][0]
    saved_base = sd2["state"]["__base_optimizer"]
    sd2["state"]["__base_optimizer"] = base2
    bde.DeepEnsemble([(model2, opt2)]).load_state_dict(saved)             # iwildcam.py:84-88
    reloaded = opt2.particles.detach().cpu().clone().numpy()
    loss2 = float(opt2.step(lambda: F.mse_loss(model2(x), y), lambda l: l.backward()))
    after2 = opt2.particles.detach().cpu().clone().numpy()
    # the pickled base optimizer carries the fused momentum in torch's own layout (resumable unfused / in the reference)
    mom = torch.cat([saved_base.state[p]["momentum_buffer"].reshape(-1).cpu() for p in saved_base.param_groups[0]["params"]])
    np.savez(os.path.join(out_dir, f"resume{rank}.npz"), loaded=loaded, after1=after1, reloaded=reloaded, after2=after2,
             losses=np.array([loss1, loss2]), n_particle_keys=np.array(len(keys)), momentum=mom.numpy())


def check(out_dir, world=2):
    ck = torch.load(os.path.join(GOLD, "ref_svgd_checkpoint4.pt"), weights_only=False)
    nxt = torch.load(os.path.join(GOLD, "ref_svgd_checkpoint4_next.pt"), weights_only=False)
    ranks = [np.load(os.path.join(out_dir, f"resume{r}.npz")) for r in range(world)]
    for r in ranks[1:]:
        for key in ranks[0].files:
            np.testing.assert_array_equal(ranks[0][key], r[key], err_msg=key)
    r0 = ranks[0]
    np.testing.assert_array_equal(r0["loaded"], ck["particles"].numpy())
    np.testing.assert_array_equal(r0["reloaded"], r0["after1"])
    assert int(r0["n_particle_keys"]) == 4                                  # every particle in every rank's dict
    for got, want, loss, want_loss in ((r0["after1"], nxt["particles_after"][0], r0["losses"][0], nxt["losses"][0]),
                                       (r0["after2"], nxt["particles_after"][1], r0["losses"][1], nxt["losses"][1])):
        np.testing.assert_allclose(got, want.numpy(), rtol=2e-5, atol=3e-6)
        assert abs(loss - want_loss) <= 2e-5 * abs(want_loss)
    assert np.abs(r0["momentum"]).max() > 0
