"""world_size-2 gloo tests of the multi-GPU paths (run on CPU with the oracle
checker backend): SVGD particle sharding with ONE all-gather of gradient rows
reproduces the single-process trajectory; MultiSWAG fan-out covers every
(member, sample) unit exactly once and is independent of the world size."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make(seed, m, ops, pg=None, fuse=False, base="sgd", tiny=False, **kw):
    import beyond_deep_ensembles_amd as bde
    torch.manual_seed(seed)
    # tiny: 14 parameters -> with two ranks the second rank's column slice holds no parameter at all
    model = nn.Sequential(nn.Linear(13, 1)) if tiny else nn.Sequential(nn.Linear(13, 20), nn.Tanh(), nn.Linear(20, 1))
    if base == "sgd":
        base = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    else:
        base = torch.optim.Adam(model.parameters(), lr=0.01)
    opt = bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model), base, particle_count=m,
                            dataset_size=64, l2_reg=0.01, process_group=pg, fuse_base_optimizer=fuse, _ops=ops, **kw)
    return model, opt


def _run_steps(model, opt, steps=3):
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(64, 13, generator=g), torch.randn(64, 1, generator=g)
    losses = []
    for t in range(steps):
        xb, yb = x[t * 16:(t + 1) * 16], y[t * 16:(t + 1) * 16]
        losses.append(float(opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())))
    return losses


import contextlib


@contextlib.contextmanager
def _backend(name):
    """"oracle": the CPU checker; "emu": the product's HipOps over the kernel sources on the CPU model (tests/hip_emu)."""
    if name == "emu":
        from tests.hip_emu.emu_ops import ALL, emulated
        with emulated(ALL) as ops:
            yield ops
    else:
        from tests.oracle_ops import OracleOps
        yield OracleOps()


def _svgd_worker(rank, world, port, m, fuse, kw, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HIP_EMU_WORKERS", "2")                  # two ranks share the container's cores
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kw = dict(kw)
    try:
        with _backend(kw.pop("backend", "oracle")) as ops:
            _svgd_rank(rank, m, fuse, kw, out_dir, ops)
    finally:
        dist.destroy_process_group()


def _svgd_rank(rank, m, fuse, kw, out_dir, ops):
    if True:
        torch.set_num_threads(1)
        # different local RNG state per rank: the constructor must still agree on the particles (broadcast)
        model, opt = _make(100 + rank, m, ops, pg=dist.group.WORLD, fuse=fuse, **dict(kw))
        fwd_calls = [0]
        orig = model.forward

        def counting_forward(*a, **k):
            fwd_calls[0] += 1
            return orig(*a, **k)
        model.forward = counting_forward
        early = []
        if dict(kw).get("overlap_backward"):
            end = opt._end_particle

            def spying_end(idx):                      # how many chunks had already left when backward returned
                if opt._ov is not None:
                    early.append(sum(w is not None for w in opt._ov["works"]))
                return end(idx)
            opt._end_particle = spying_end
        losses = _run_steps(model, opt)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), particles=opt.particles.numpy(), losses=np.array(losses),
                 fwd=np.array(fwd_calls[0]), early=np.array(early))


@pytest.mark.parametrize("m,fuse,kw", [
    (4, False, {}), (2, True, {}),
    (4, False, {"exchange_chunks": 3}), (4, True, {"exchange_chunks": 2}), (2, True, {"exchange_chunks": 4, "base": "adam"}),
    (4, True, {"exchange": "alltoall"}), (2, True, {"exchange": "alltoall"}), (4, True, {"exchange": "alltoall", "base": "adam"}),
    (2, True, {"exchange": "alltoall", "tiny": True}), (4, False, {"exchange_chunks": 7, "tiny": True}),
    (4, True, {"exchange_chunks": 3, "overlap_backward": True}), (2, False, {"exchange_chunks": 4, "overlap_backward": True}),
    (2, True, {"exchange_chunks": 5, "overlap_backward": True, "base": "adam"}),
    (20, True, {"base": "adam"}),
], ids=["allgather", "allgather_fused", "pipelined", "pipelined_fused", "pipelined_fused_adam", "alltoall_2per",
        "alltoall_1per", "alltoall_adam", "alltoall_empty_slice", "pipelined_more_chunks_than_columns",
        "overlap_fused_2per", "overlap_unfused_1per", "overlap_fused_adam_1per", "allgather_20_particles_one_apply_launch"])
def test_svgd_sharded_equals_single_process(tmp_path, m, fuse, kw):
    """One exchange of gradient rows (all-gather, chunk-pipelined all-gather, or the dimension-sharded all-to-all
    pair) + the deterministic update reproduces the single-process trajectory, with identical particles on all ranks."""
    from tests.oracle_ops import OracleOps
    world = 2
    port = _free_port()
    mp.spawn(_svgd_worker, args=(world, port, m, fuse, tuple(kw.items()), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # replicas stay bit-identical across ranks
    np.testing.assert_array_equal(r0["particles"], r1["particles"])
    np.testing.assert_array_equal(r0["losses"], r1["losses"])
    # each rank ran forward/backward only for its own M/W particles
    assert int(r0["fwd"]) == 3 * m // world and int(r1["fwd"]) == 3 * m // world
    if kw.get("overlap_backward"):
        # chunk gathers left WHILE the last local particle's backward pass was running (every step, both ranks)
        for r in (r0, r1):
            assert len(r["early"]) == 3 and all(int(n) >= 1 for n in r["early"]), r["early"]
    # single-process run from rank 0's initial state
    torch.set_num_threads(1)
    model, opt = _make(100, m, OracleOps(), fuse=fuse, base=kw.get("base", "sgd"), tiny=kw.get("tiny", False))
    losses = _run_steps(model, opt)
    np.testing.assert_allclose(r0["particles"], opt.particles.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(r0["losses"], np.array(losses), rtol=1e-6)


@pytest.mark.parametrize("world,fuse,kw", [
    (8, True, {}), (8, True, {"exchange_chunks": 3}), (8, True, {"exchange": "alltoall"}), (8, False, {}),
    (4, True, {"exchange_chunks": 2, "overlap_backward": True}), (4, True, {"exchange": "alltoall", "base": "adam"}),
], ids=["8_ranks_allgather_fused", "8_ranks_pipelined_fused", "8_ranks_alltoall", "8_ranks_allgather_torch_loop",
        "4_ranks_overlap_fused", "4_ranks_alltoall_adam"])
def test_svgd_eight_particles_sharded_over_four_and_eight_ranks(tmp_path, world, fuse, kw):
    """BASELINE configs[3] / north_star: 8 particles sharded ONE per rank over 8 ranks (two per rank over 4), every exchange
    mode -- the all-gather of the gradient rows, its chunk pipeline, the dimension-sharded all-to-all pair -- over gloo: every
    rank runs forward / backward for its own particle(s) only, all ranks end every step with bit-identical particles and
    losses, and the trajectory is the single-process one.  (What the driver's 8-GPU scaling run executes over RCCL.)"""
    from tests.oracle_ops import OracleOps
    m = 8
    mp.spawn(_svgd_worker, args=(world, _free_port(), m, fuse, tuple(kw.items()), str(tmp_path)), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in ranks[1:]:
        np.testing.assert_array_equal(ranks[0]["particles"], r["particles"])
        np.testing.assert_array_equal(ranks[0]["losses"], r["losses"])
    assert all(int(r["fwd"]) == 3 * m // world for r in ranks)
    torch.set_num_threads(1)
    model, opt = _make(100, m, OracleOps(), fuse=fuse, base=kw.get("base", "sgd"))
    losses = _run_steps(model, opt)
    np.testing.assert_allclose(ranks[0]["particles"], opt.particles.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ranks[0]["losses"], np.array(losses), rtol=1e-6)


def _forced_worker(rank, world, port, m, fuse, kw, out_dir):
    _svgd_worker(rank, world, port, m, fuse, kw, out_dir)


@pytest.mark.parametrize("m,fuse,kw", [
    (4, False, {}), (4, True, {}), (4, True, {"exchange_chunks": 3}), (2, True, {"exchange_chunks": 3, "overlap_backward": True}),
    (4, True, {"exchange": "alltoall"}), (2, True, {"exchange": "alltoall", "base": "adam"}),
], ids=["allgather", "allgather_fused", "pipelined_fused", "overlap_fused", "alltoall", "alltoall_adam"])
def test_svgd_group_of_one_rank_with_forced_exchange(tmp_path, m, fuse, kw):
    """``_force_exchange=True``: a process group of ONE rank still goes through the collectives (the switch that lets a
    single-GPU box execute the RCCL branches, tests/test_dist_gpu.py); the result is the single-process trajectory."""
    from tests.oracle_ops import OracleOps
    forced = dict(kw, _force_exchange=True)
    mp.spawn(_forced_worker, args=(1, _free_port(), m, fuse, tuple(forced.items()), str(tmp_path)), nprocs=1, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    assert int(r0["fwd"]) == 3 * m
    if kw.get("overlap_backward"):
        assert len(r0["early"]) == 3 and all(int(n) >= 1 for n in r0["early"])
    torch.set_num_threads(1)
    model, opt = _make(100, m, OracleOps(), fuse=fuse, base=kw.get("base", "sgd"))
    losses = _run_steps(model, opt)
    np.testing.assert_allclose(r0["particles"], opt.particles.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(r0["losses"], np.array(losses), rtol=1e-6)


@pytest.mark.parametrize("m,fuse,kw", [
    (4, True, {}), (2, True, {"exchange_chunks": 3, "overlap_backward": True}),
    (4, True, {"exchange": "alltoall"}), (4, True, {"exchange": "alltoall", "base": "adam"}),
], ids=["allgather_fused", "overlap_fused", "alltoall", "alltoall_adam"])
def test_svgd_two_ranks_on_the_cpu_model(tmp_path, m, fuse, kw):
    """The same two-rank runs with the REAL kernels in both ranks -- HipOps over the kernel sources on the CPU model of
    tests/hip_emu (column-slice Gram + fp64 Gram blocks + statistics from the ranks' blocks for the dimension-sharded
    exchange, the fused update on a slice, the packer of the pipelined all-gather): replicas bit-identical, and equal to the
    single-process run of the same kernels (what tests/test_dist_gpu.py checks with several ranks on one MI355X)."""
    from tests.hip_emu import build
    if not build.available():
        pytest.skip("no host clang / HIP headers to build the CPU model with")
    build.build(__import__("tests.hip_emu.emu_ops", fromlist=["ALL"]).ALL)          # once, before two ranks race for it
    world = 2
    mp.spawn(_svgd_worker, args=(world, _free_port(), m, fuse, tuple(dict(kw, backend="emu").items()), str(tmp_path)),
             nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["particles"], r1["particles"])
    np.testing.assert_array_equal(r0["losses"], r1["losses"])
    torch.set_num_threads(1)
    with _backend("emu") as ops:
        model, opt = _make(100, m, ops, fuse=fuse, base=kw.get("base", "sgd"), single_launch=False)
        losses = _run_steps(model, opt)
        particles = opt.particles.numpy()
    # the sharded Gram sums per-rank fp64 blocks, the single process one fp32-partial reduction: 1e-5 as on the device
    np.testing.assert_allclose(r0["particles"], particles, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r0["losses"], np.array(losses), rtol=1e-6)


def _overlap_edit_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.oracle_ops import OracleOps
        torch.set_num_threads(1)
        model, opt = _make(100 + rank, 2, OracleOps(), pg=dist.group.WORLD, fuse=True, exchange_chunks=3, overlap_backward=True)
        x, y = torch.randn(16, 13), torch.randn(16, 1)

        def backward_then_clip(loss):
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1e-3)      # edits the gradients behind the exchange
        try:
            opt.step(lambda: F.mse_loss(model(x), y), backward_then_clip)
            msg = "no error"
        except RuntimeError as e:
            msg = str(e)
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write(msg)
    finally:
        dist.destroy_process_group()


def test_overlap_backward_refuses_gradients_edited_after_backward(tmp_path):
    """with overlap_backward the last local particle's chunks leave inside backward(); a closure that clips /
    scales / accumulates afterwards would change the gradients behind the exchange -> a clear error on every rank."""
    mp.spawn(_overlap_edit_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        msg = (tmp_path / f"rank{r}.txt").read_text()
        assert "changed after its column chunk had been sent" in msg, msg


def _swag_member(seed, ops):
    import beyond_deep_ensembles_amd as bde
    torch.manual_seed(seed)
    model = nn.Linear(6, 2)
    opt = bde.SwagOptimizer(model.parameters(), torch.optim.SGD(model.parameters(), lr=0.1), update_interval=1,
                            deviation_samples=4, rng="philox", seed=seed, _ops=ops)
    x = torch.randn(8, 6)
    for _ in range(6):
        opt.step(lambda: model(x).pow(2).mean(), lambda l: l.backward())
    return model, opt


def test_multiswag_fan_out_is_world_size_independent():
    """MultiSWAG (BASELINE config 5 shape: members x samples fanned over GPUs):
    the union of the ranks' outputs, put back in unit order, does not depend
    on the number of ranks."""
    import beyond_deep_ensembles_amd as bde
    from beyond_deep_ensembles_amd.ensemble import fan_out
    from tests.oracle_ops import OracleOps
    ops = OracleOps()
    samples, members = 13, 3
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(1))

    def run(world):
        outs = {}
        for rank in range(world):
            ens = bde.DeepEnsemble([_swag_member(10 + i, ops) for i in range(members)])
            res = ens.predict(lambda m: m(x).detach(), samples, rank=rank, world_size=world)
            for (u, _, _), o in zip(fan_out(samples, members, rank, world), res):
                outs[u] = o
        return torch.stack([outs[u] for u in range(samples)])

    a, b, c = run(2), run(3), run(8)
    assert torch.equal(a, b) and torch.equal(a, c)


def _predict_worker(rank, world, port, out_dir, members=3, samples=13):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import beyond_deep_ensembles_amd as bde
        from tests.oracle_ops import OracleOps
        torch.set_num_threads(1)
        ops = OracleOps()
        ens = bde.DeepEnsemble([_swag_member(10 + i, ops) for i in range(members)])
        x = torch.randn(4, 6, generator=torch.Generator().manual_seed(1))
        out = ens.predict_distributed(lambda m: m(x).detach(), samples, dist.group.WORLD)
        np.save(os.path.join(out_dir, f"pred{rank}.npy"), out.numpy())
    finally:
        dist.destroy_process_group()


def test_multiswag_five_modes_thirty_samples_over_eight_ranks(tmp_path):
    """BASELINE configs[4]: MultiSWAG with 5 modes x 30 posterior samples = 150 (member, sample) units fanned over 8 ranks
    (DeepEnsemble.predict_distributed over gloo): every rank ends with the full [150, ...] prediction tensor, identical on all
    ranks and identical, unit for unit, to the single-process DeepEnsemble.predict (ensemble.py:28-44: 30 samples per member, in
    member order); the 8 ranks' unit ranges are contiguous, cover 150 exactly once and differ in size by at most one."""
    import beyond_deep_ensembles_amd as bde
    from beyond_deep_ensembles_amd.ensemble import fan_out
    from tests.oracle_ops import OracleOps
    world, members, samples = 8, 5, 150
    mp.spawn(_predict_worker, args=(world, _free_port(), str(tmp_path), members, samples), nprocs=world, join=True)
    preds = [np.load(tmp_path / f"pred{r}.npy") for r in range(world)]
    for p in preds[1:]:
        np.testing.assert_array_equal(preds[0], p)
    assert preds[0].shape[0] == samples
    units = [[u for u, _, _ in fan_out(samples, members, r, world)] for r in range(world)]
    assert sorted(u for us in units for u in us) == list(range(samples))
    assert all(us == list(range(us[0], us[0] + len(us))) for us in units) and {len(us) for us in units} <= {18, 19}
    ens = bde.DeepEnsemble([_swag_member(10 + i, OracleOps()) for i in range(members)])
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(1))
    want = ens.predict(lambda m: m(x).detach(), samples)
    np.testing.assert_array_equal(preds[0], torch.stack(list(want)).numpy() if not torch.is_tensor(want) else want.numpy())


def test_predict_distributed_gathers_in_reference_order(tmp_path):
    """DeepEnsemble.predict_distributed: every rank ends with the full [S, ...] tensor, identical to the
    single-process fan-out put back in unit order."""
    import beyond_deep_ensembles_amd as bde
    from beyond_deep_ensembles_amd.ensemble import fan_out
    from tests.oracle_ops import OracleOps
    world = 2
    mp.spawn(_predict_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "pred0.npy"), np.load(tmp_path / "pred1.npy")
    np.testing.assert_array_equal(a, b)
    assert a.shape[0] == 13
    ops = OracleOps()
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(1))
    outs = {}
    for rank in range(world):
        ens = bde.DeepEnsemble([_swag_member(10 + i, ops) for i in range(3)])
        res = ens.predict(lambda m: m(x).detach(), 13, rank=rank, world_size=world)
        for (u, _, _), o in zip(fan_out(13, 3, rank, world), res):
            outs[u] = o
    want = torch.stack([outs[u] for u in range(13)]).numpy()
    np.testing.assert_array_equal(a, want)


def _svgd_member(seed, ops):
    import beyond_deep_ensembles_amd as bde
    torch.manual_seed(seed)
    model = nn.Linear(6, 2)
    base = torch.optim.SGD(model.parameters(), lr=0.1)
    opt = bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model), base, particle_count=4,
                            dataset_size=8, _ops=ops)
    x = torch.randn(8, 6)
    opt.step(lambda: model(x).pow(2).mean(), lambda l: l.backward())
    return model, opt


@pytest.mark.parametrize("kind", ["svgd", "swag"])
def test_fan_out_reproduces_single_process_predict(kind):
    """The fan-out must evaluate exactly the (member, sample) units of the reference's sequential loop
    (ensemble.py:37-44): SVGD members cycle through their particles from where the single-process call would be,
    SWAG members use the Philox stream of the sample's index -- and afterwards every sampler stands where the
    single-process call leaves it, on every rank."""
    import beyond_deep_ensembles_amd as bde
    from beyond_deep_ensembles_amd.ensemble import fan_out
    from tests.oracle_ops import OracleOps
    ops = OracleOps()
    make = _svgd_member if kind == "svgd" else _swag_member
    samples, members = 11, 2                      # split 6 + 5: more samples than particles, so the cycle wraps
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(1))
    ens = bde.DeepEnsemble([make(10 + i, ops) for i in range(members)])
    want = ens.predict(lambda m: m(x).detach().clone(), samples)
    want2 = ens.predict(lambda m: m(x).detach().clone(), samples)          # a second call continues the sequences
    for world in (2, 3, 4):
        outs, outs2 = {}, {}
        for rank in range(world):
            ens_r = bde.DeepEnsemble([make(10 + i, ops) for i in range(members)])
            for store in (outs, outs2):
                res = ens_r.predict(lambda m: m(x).detach().clone(), samples, rank=rank, world_size=world)
                for (u, _, _), o in zip(fan_out(samples, members, rank, world), res):
                    store[u] = o
        assert torch.equal(torch.stack([outs[u] for u in range(samples)]), want), (kind, world)
        assert torch.equal(torch.stack([outs2[u] for u in range(samples)]), want2), (kind, world)
    if kind == "svgd":
        assert not torch.equal(want[0], want[1])                           # different particles really differ


# ------------------------------------------------------------------ checkpoints in every exchange mode --
def _resume_worker(rank, world, port, kw, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.ckpt_resume import resume_worker
        from tests.oracle_ops import OracleOps
        torch.set_num_threads(1)
        resume_worker(rank, OracleOps(), torch.device("cpu"), dist.group.WORLD, dict(kw), out_dir)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kw", [{}, {"exchange_chunks": 3}, {"exchange": "alltoall"}], ids=["allgather", "pipelined", "alltoall"])
def test_reference_checkpoint_resumes_on_two_ranks(tmp_path, kw):
    """tests/ckpt_resume.py: a reference-written SVGD checkpoint -> 2-rank optimizer -> step -> own state_dict() (a
    collective in alltoall mode) -> fresh optimizer -> step; both steps == the reference's own next steps."""
    from tests.ckpt_resume import check
    mp.spawn(_resume_worker, args=(2, _free_port(), tuple(kw.items()), str(tmp_path)), nprocs=2, join=True)
    check(str(tmp_path))


# ------------------------------------------------------------------ the RCCL one-rank matrix of tests/test_dist_gpu.py, on the CPU model --
def _one_rank_matrix_worker(rank, world, port, m, kw, out_dir):
    import tests.test_dist_gpu as G
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        with _backend("emu") as ops:
            model, opt = G._make_svgd(100 + rank, m, torch.device("cpu"), pg=dist.group.WORLD, _ops=ops, **dict(kw))
            losses = G._run_steps(model, opt, torch.device("cpu"))
            np.savez(os.path.join(out_dir, f"rank{rank}.npz"), particles=opt.particles.numpy(), losses=np.array(losses))
    finally:
        dist.destroy_process_group()


def _svgd_cases():
    import tests.test_dist_gpu as G
    return G.SVGD_CASES


@pytest.mark.parametrize("name,m,kw", _svgd_cases(), ids=[c[0] for c in _svgd_cases()])
def test_one_rank_forced_exchange_matrix_on_the_cpu_model(tmp_path, name, m, kw):
    """tests/test_dist_gpu.py::test_svgd_rccl_one_rank_forced_exchange -- the nine exchange variants through a process group of
    ONE rank with the collectives forced -- has run on the MI355X for one variant only (the pool closed).  Here the same nine
    variants, the same model, optimizer and steps (its _make_svgd / _run_steps), the product's kernels on the CPU model, gloo
    instead of RCCL: the result must equal the run without a process group exactly as the GPU test demands -- bit for bit for
    the replicated exchanges, to 1e-5 / 2e-7 for the dimension-sharded exchange and reuse_gram, whose Gram partials are summed
    in another fixed order; that difference is printed (max 2.98e-08 = one ulp at the particles' scale for fused_sgd_reuse, the same
    maximum the MI355X showed in round 4, profiles/r04_pytest_gpu_call_b.log; which elements differ depends on expf, host vs device)."""
    import tests.test_dist_gpu as G
    from tests.hip_emu import build
    if not build.available():
        pytest.skip("no host clang / HIP headers to build the CPU model with")
    build.build(__import__("tests.hip_emu.emu_ops", fromlist=["ALL"]).ALL)
    forced = dict(kw, _force_exchange=True)
    mp.spawn(_one_rank_matrix_worker, args=(1, _free_port(), m, tuple(forced.items()), str(tmp_path)), nprocs=1, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    single_kw = {k: v for k, v in kw.items() if k not in ("exchange", "exchange_chunks", "overlap_backward")}
    torch.set_num_threads(1)
    with _backend("emu") as ops:
        model, opt = G._make_svgd(100, m, torch.device("cpu"), single_launch=False, _ops=ops, **single_kw)
        losses = G._run_steps(model, opt, torch.device("cpu"))
        want = opt.particles.numpy().copy()
    if kw.get("exchange") == "alltoall" or kw.get("reuse_gram"):
        a, b = r0["particles"].astype(np.float32), want.astype(np.float32)
        ulp = np.abs(a - b) / np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32))
        print(f"one_rank_matrix[{name}] on the CPU model: max |sharded - single| = {np.abs(a - b).max():.3e} = {ulp.max():.1f} ulp")
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=2e-7)
        np.testing.assert_allclose(r0["losses"], np.array(losses), rtol=2e-6)
    else:
        np.testing.assert_array_equal(r0["particles"], want)
        np.testing.assert_array_equal(r0["losses"], np.array(losses))
