"""BASELINE configs[3] and configs[4] in their multi-rank form AT THEIR REAL SIZE, on the one GPU a test box has:

* configs[3] -- iWildCam ResNet-50 SVGD, 8 particles x D = 23,880,950, sharded over 2 and over 8 ranks (all ranks on
  ``cuda:0``, gloo carrying the CUDA tensors), each of the three exchanges (``allgather`` -- what north_star names --,
  the chunk-pipelined all-gather with 8 chunks, the dimension-sharded ``alltoall``).  Two full posterior updates
  (``SVGDOptimizer._posterior_update``: exchange, statistics, -phi, 8 shared-state SGD applications) must leave
  bit-identical replicas on every rank, equal the single-process HIP update, and match the CPU oracle
  (``oracle.svgd_phi_cols`` + ``oracle.svgd_apply_shared_optimizer``, fp64-anchored) on the first 2^20 and the last
  1,003 columns -- the slices ``test_fullsize_gpu.py`` uses.  Reference semantics: ``src/algos/svgd.py:65-105``.
* configs[4] -- Camelyon17 DenseNet-121 MultiSWAG, 5 members x D = 6,955,906 (364 tensors each), 150 posterior
  samples fanned over 8 ranks (``DeepEnsemble.predict_distributed``): at most two members per rank, one batched
  sampling pass per member block, output == the single-process ``predict`` (the reference's order,
  ``src/algos/ensemble.py:28-44``) and == the oracle's ``swag_sample`` with the Philox noise of the unit's stream.

The children are fresh ``torch.multiprocessing.spawn`` processes; the pytest process is never re-exec'd.
"""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.spawn_one_device import spawn_ranks

pytestmark = pytest.mark.gpu

M = 8
D50 = 23_880_950            # torchvision ResNet-50 + 182-class head
D121 = 6_955_906            # Camelyon DenseNet-121
HEAD = 372_918
L2, SCALE, N_DATA = 1e-5, 1.0, 129_809.0
SGD = dict(lr=0.5, momentum=0.9, nesterov=True, weight_decay=3e-4)
SLICES = (slice(0, 1 << 20), slice(D50 - 1003, D50))
MODES = {"allgather": {}, "pipelined8": {"exchange_chunks": 8}, "alltoall": {"exchange": "alltoall"}}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=8))
    return dist, torch.device("cuda", 0)


# ------------------------------------------------------------------------------------------------ SVGD, configs[3]
def _particle_rows(dev):
    """The 8 initial particles: shared backbone, head-only re-initialisation (iwildcam/models.py:118-119)."""
    g = torch.Generator(device=dev).manual_seed(1234)
    theta0 = torch.randn(D50, device=dev, generator=g) * 0.05
    rows = []
    for _ in range(M):
        r = theta0.clone()
        r[D50 - HEAD:] += (torch.rand(HEAD, device=dev, generator=g) * 2 - 1) / math.sqrt(2048)
        rows.append(r)
    return rows


def _grad_row(dev, step, particle):
    g = torch.Generator(device=dev).manual_seed(5000 + 16 * step + particle)
    return torch.randn(D50, device=dev, generator=g) * 0.01


def _build(dev, pg=None, **kw):
    import beyond_deep_ensembles_amd as bde
    rows = _particle_rows(dev)
    theta = torch.nn.Parameter(rows[0].clone())             # ONE flat parameter: no model, the update is the subject
    nxt = iter(range(1, M))

    def reset():
        with torch.no_grad():
            theta.copy_(rows[next(nxt)])
    base = torch.optim.SGD([theta], **SGD)
    opt = bde.SVGDOptimizer([theta], reset, base, particle_count=M, dataset_size=N_DATA, l2_reg=L2,
                            kernel_grad_scale=SCALE, process_group=pg, fuse_base_optimizer=True, **kw)
    del rows
    return opt


def _two_updates(opt, dev, rank, world):
    """Drive the update exactly as SVGDOptimizer.step does after its backward passes; returns what the parent compares."""
    per = M // world
    own = range(rank * per, (rank + 1) * per)
    out = {}
    for step in range(2):
        loss_sum = torch.zeros((), device=dev)
        for i in own:
            opt._grad_row(i)[:D50] = _grad_row(dev, step, i)
            loss_sum += 0.1 * (i + 1) + step
        loss = opt._posterior_update(loss_sum)
        torch.cuda.synchronize()
        P = opt.particles                                       # a collective in alltoall mode: every rank reads it
        out[f"loss{step}"] = np.array(float(loss))
        out[f"bits{step}"] = np.array(int(P.contiguous().view(torch.int32).to(torch.int64).sum().item()))
        for k, sl in enumerate(SLICES):
            out[f"p{step}_{k}"] = P[:, sl].cpu().numpy()
        del P
    return out


def _svgd_worker(rank, world, port, mode, out_dir):
    dist, dev = _init(rank, world, port)
    try:
        torch.manual_seed(100 + rank)                           # local RNG state differs per rank on purpose
        opt = _build(dev, pg=dist.group.WORLD, **MODES[mode])
        res = _two_updates(opt, dev, rank, world)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def single_process():
    """The same two updates without a process group (same kernels), and the oracle's first update on the slices."""
    from oracle import bde_oracle as O
    dev = torch.device("cuda", 0)
    rows = torch.stack([r for r in _particle_rows(dev)])
    P0 = rows.cpu()
    del rows
    G0 = torch.stack([_grad_row(dev, 0, i).cpu() for i in range(M)])
    oracle = []
    d2_32, d2_64 = O.svgd_sq_dists(P0), O.svgd_sq_dists(P0.double())            # full width, once
    for sl in SLICES:
        upd = []
        for dt, d2 in ((torch.float32, d2_32), (torch.float64, d2_64)):
            Ps, Gs = P0[:, sl].to(dt), G0[:, sl].to(dt)
            phi = O.svgd_phi_cols(Ps, Gs, d2, L2, SCALE, N_DATA)
            part = [[Ps[i].clone()] for i in range(M)]
            param = torch.nn.Parameter(torch.zeros(Ps.shape[1], dtype=dt))
            O.svgd_apply_shared_optimizer(part, [[-phi[i]] for i in range(M)], [param], torch.optim.SGD([param], **SGD))
            upd.append(torch.stack([p[0] for p in part]))
        oracle.append((upd[0], upd[1], P0[:, sl].double()))
    del P0, G0
    opt = _build(dev)
    res = _two_updates(opt, dev, 0, 1)
    del opt
    torch.cuda.empty_cache()
    return res, oracle


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("mode", list(MODES))
def test_svgd_resnet50_sharded_one_device(tmp_path, single_process, mode, world):
    single, oracle = single_process
    spawn_ranks(_svgd_worker, lambda: (world, _free_port(), mode, str(tmp_path)), world)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    r0 = ranks[0]
    for r in ranks[1:]:                                          # replicas bit-identical across ranks (whole matrix)
        for key in r0.files:
            np.testing.assert_array_equal(r0[key], r[key], err_msg=key)
    for step in range(2):
        want_loss = sum(0.1 * (i + 1) + step for i in range(M)) / M
        assert abs(float(r0[f"loss{step}"]) - want_loss) <= 1e-5 * want_loss
        for k in range(len(SLICES)):
            ours, one = r0[f"p{step}_{k}"], single[f"p{step}_{k}"]
            if mode != "alltoall":
                # same kernels, same reduction order: the replicated exchanges reproduce the single-process bits
                np.testing.assert_array_equal(ours, one)
            else:
                # slice-wise Gram partials are summed in a different (fixed) order: fp32-rounding-level differences
                np.testing.assert_allclose(ours, one, rtol=0, atol=2e-7 * (step + 1))
    # first update vs the CPU oracle, fp64-anchored: |ours - fp64| <= max(2 |oracle32 - fp64|, 3e-6 max|update|)
    for k, (u32, u64, p0) in enumerate(oracle):
        ours = torch.from_numpy(r0[f"p0_{k}"]).double()
        err = (ours - u64).abs().max().item()
        err_ref = (u32.double() - u64).abs().max().item()
        mag = (u64 - p0).abs().max().item()
        assert err <= max(2 * err_ref, 3e-6 * mag), (mode, world, k, err, err_ref, mag)


# -------------------------------------------------------------------------------------------- MultiSWAG, configs[4]
K, SAMPLES, MEMBERS, TENSORS = 20, 150, 5, 364
PROBE_HEAD, PROBE_TAIL = 4096, 1003


def _member(dev, index):
    import beyond_deep_ensembles_amd as bde
    g = torch.Generator(device=dev).manual_seed(700 + index)
    sizes = [D121 // TENSORS] * (TENSORS - 1)
    sizes.append(D121 - sum(sizes))
    params = [torch.nn.Parameter(torch.randn(s, device=dev, generator=g) * 0.05) for s in sizes]
    opt = bde.SwagOptimizer(params, torch.optim.SGD(params, lr=1e-3), update_interval=1, deviation_samples=K,
                            rng="philox", seed=900 + index)
    with torch.no_grad():
        for _ in range(K + 3):                                   # every ring row written, head wrapped
            opt._theta[:D121] += torch.randn(D121, device=dev, generator=g) * 1e-3
            opt._swag_update()
    model = torch.nn.Module()
    model.p = torch.nn.ParameterList(params)
    return model, opt


def _probe(model):
    """A prediction that exposes the sampled weights themselves: head of the first tensor + the last PROBE_TAIL weights."""
    ps = list(model.p)
    tail, need = [], PROBE_TAIL
    for p in reversed(ps):
        v = p.detach().reshape(-1)
        tail.insert(0, v[-need:] if v.numel() >= need else v)
        need -= min(need, v.numel())
        if need == 0:
            break
    return torch.cat([ps[0].detach().reshape(-1)[:PROBE_HEAD]] + tail)


def _predict_worker(rank, world, port, out_dir):
    dist, dev = _init(rank, world, port)
    try:
        import beyond_deep_ensembles_amd as bde
        from beyond_deep_ensembles_amd.ensemble import members_needed
        need = members_needed(SAMPLES, MEMBERS, rank, world)
        # a rank only ever samples from the members its unit range touches: build those, placeholders for the rest
        pairs = []
        for i in range(MEMBERS):
            pairs.append(_member(dev, i) if i in need else _placeholder(dev, i))
        ens = bde.DeepEnsemble(pairs)
        calls = [0]
        for i in need:
            opt = ens.optimizers[i]
            orig = opt._ops.swag_sample_batched

            def counted(*a, _orig=orig, **k):
                calls[0] += 1
                return _orig(*a, **k)
            opt._ops = _Counting(opt._ops, counted)
        out = ens.predict_distributed(_probe, SAMPLES, dist.group.WORLD)
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"pred{rank}.npz"), out=out.cpu().numpy(), batched=np.array(calls[0]),
                 need=np.array(need))
    finally:
        dist.destroy_process_group()


class _Counting:
    def __init__(self, real, counted):
        self._real, self.swag_sample_batched = real, counted

    def __getattr__(self, name):
        return getattr(self._real, name)


def _placeholder(dev, index):
    """A member this rank never samples from: a SWAG optimizer over a tiny model, so that the ensemble keeps its
    5-member shape (the sample split depends on the member count) without its 670 MB of statistics."""
    import beyond_deep_ensembles_amd as bde
    params = [torch.nn.Parameter(torch.zeros(8, device=dev))]
    opt = bde.SwagOptimizer(params, torch.optim.SGD(params, lr=1e-3), update_interval=1, deviation_samples=K,
                            rng="philox", seed=900 + index)
    model = torch.nn.Module()
    model.p = torch.nn.ParameterList(params)
    return model, opt


def test_multiswag_densenet121_fanout_eight_ranks_one_device(tmp_path):
    import beyond_deep_ensembles_amd as bde
    from beyond_deep_ensembles_amd.ensemble import fan_out, members_needed, split_samples
    from oracle import bde_oracle as O
    from oracle import philox as PH
    world = 8
    spawn_ranks(_predict_worker, lambda: (world, _free_port(), str(tmp_path)), world)
    preds = [np.load(tmp_path / f"pred{r}.npz") for r in range(world)]
    for r, p in enumerate(preds):
        np.testing.assert_array_equal(preds[0]["out"], p["out"])
        need = members_needed(SAMPLES, MEMBERS, r, world)
        assert len(need) <= 2 and list(p["need"]) == need
        assert int(p["batched"]) == len(need), (r, int(p["batched"]), need)      # ONE batched pass per member block
    got = preds[0]["out"]
    assert got.shape == (SAMPLES, PROBE_HEAD + PROBE_TAIL)
    # the single-process predict (the reference's order: member 0's samples, then member 1's, ...)
    dev = torch.device("cuda", 0)
    ens = bde.DeepEnsemble([_member(dev, i) for i in range(MEMBERS)])
    want = ens.predict(_probe, SAMPLES).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    # the oracle's sample (swag.py:57,112-114) with the Philox noise of the unit's stream, on the probed columns
    counts = split_samples(SAMPLES, MEMBERS)
    starts = np.cumsum([0] + counts[:-1])
    g_tail0 = (D121 - PROBE_TAIL) // 4
    for unit in (0, 29, 30, 77, 149):
        member = int(np.searchsorted(starts, unit, side="right") - 1)
        s = unit - int(starts[member])
        opt = ens.optimizers[member]
        rd = PH.SWAG_ROUNDS                                             # the samplers' noise streams
        eps_w = torch.from_numpy(PH.normals(opt.seed, s, K, PH.DOMAIN_LOWRANK, rounds=rd)).float()
        head = PH.box_muller(PH.stream_bits(opt.seed, s, PROBE_HEAD // 4, rounds=rd)).reshape(-1)
        tail = PH.box_muller(PH.stream_bits(opt.seed, s, (D121 + 3) // 4 - g_tail0, idx0=g_tail0, rounds=rd)).reshape(-1)
        tail = tail[(D121 - PROBE_TAIL) - 4 * g_tail0:][:PROBE_TAIL]
        mean, sq, dk = opt.mean_vector(), opt.sq_vector(), opt.deviations_dk()
        for sl, eps, cols in ((slice(0, PROBE_HEAD), head, slice(0, PROBE_HEAD)),
                              (slice(D121 - PROBE_TAIL, D121), tail, slice(PROBE_HEAD, None))):
            ref = O.swag_sample(mean[sl].cpu(), sq[sl].cpu(), dk[sl].cpu(), eps_w, torch.from_numpy(eps).float())
            assert torch.allclose(torch.from_numpy(got[unit, cols]), ref, rtol=2e-5, atol=2e-6), (unit, member, s)
    assert sorted(u for r in range(world) for u, _, _ in fan_out(SAMPLES, MEMBERS, r, world)) == list(range(SAMPLES))
