"""The optimizer shells (the drop-in boundary) replayed against the golden
trajectories captured from the imported reference.

Each test runs twice:
  * backend "oracle" (CPU, `-m "not gpu"`): the shells' HOST logic -- flat
    layouts, particle/ring indexing, schedules, gradient hand-over, state keys
    -- with the arithmetic done by the CPU oracle (tests/oracle_ops.py);
  * backend "hip" (`-m gpu`): the product path, libbde_hip.so on cuda:0.
"""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import beyond_deep_ensembles_amd as bde
from tests.oracle_ops import OracleOps


@pytest.fixture(params=["oracle", "emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    """"oracle": the CPU checker behind the shells; "hip": libbde_hip.so on the MI355X; "emu": the product's HipOps over the
    kernel SOURCES compiled for the CPU execution model of tests/hip_emu (the shells then drive the real planners, C-ABI
    entry points, C++ autograd nodes and kernels, lane by lane, on CPU tensors)."""
    if request.param == "oracle":
        yield OracleOps(), torch.device("cpu")
        return
    if request.param == "emu":
        from tests.hip_emu import build, emu_ops
        if not build.available():
            pytest.skip("no host clang / HIP headers to build the CPU model with")
        import beyond_deep_ensembles_amd.bbb_layers as L
        # as on the device, the Bayesian layers go through the C++ autograd nodes (csrc/host_autograd.cpp, here compiled
        # over the CPU model; lib/_bde_host.so binds the device library)
        native = build.load_host_nodes(emu_ops.ALL)
        from beyond_deep_ensembles_amd.ops import HipOps
        monkeypatch.setattr(L, "_native_nodes", lambda ops: native if isinstance(ops, HipOps) else None)
        with emu_ops.emulated(emu_ops.ALL) as ops:
            yield ops, torch.device("cpu")
        return
    from beyond_deep_ensembles_amd.ops import HipOps
    yield HipOps(), torch.device("cuda:0")


# Code that has not been green on an MI355X for the sources in this tree is reached only by variants that carry this
# marker (collected behind every verified test: tests/conftest.py); the unmarked variants run what a default-constructed
# object runs.  The CPU-model runs of these tests enforce the split (conftest._kernel_reach_guard).
unverified = pytest.mark.device_unverified
SMALL = dict(single_launch="two", host_fast_paths=True)      # SVGDOptimizer: the small-model kernel + the round-5 native host paths


@pytest.fixture(params=["stock_conv", pytest.param("fused_conv", marks=unverified("conv_lrt"))])
def conv_kw(request):
    """BBBConv2d keyword arguments: the default (fused_conv="auto" -> the stock convolutions + fused element-wise passes while
    conv_profit.json is empty) / the fused convolution kernels forced on."""
    return {} if request.param == "stock_conv" else {"fused_conv": True}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_mlp():
    return nn.Sequential(nn.Linear(13, 50), nn.ReLU(), nn.Linear(50, 1))


def set_flat(params, flat):
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            off += n


def flat(ts):
    return torch.cat([t.detach().reshape(-1) for t in ts])


# ------------------------------------------------------------------ SVGD --
@pytest.mark.parametrize("name,make_opt,fuse", [
    ("sgd", lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4), False),
    ("adam", lambda ps: torch.optim.Adam(ps, lr=1e-3), False),
    ("adam_wd", lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-2), False),
    ("sgd", lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4), True),
    ("adam", lambda ps: torch.optim.Adam(ps, lr=1e-3), True),
    ("adam_wd", lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-2), "reuse_gram"),
    ("sgd", lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4), "reuse_gram"),
    ("adam", lambda ps: torch.optim.Adam(ps, lr=1e-3), "reuse_gram"),
    # the small-model kernel (bde_svgd_step_small*: the whole update in two launches) + the round-5 native host paths
    # instead of the streaming kernels a default-constructed optimizer takes until they are device-verified
    pytest.param("sgd", lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4), "small", marks=unverified("svgd_small", "small_step_host", "mean_scalars", "fast_loop")),
    pytest.param("adam_wd", lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-2), "small_fused", marks=unverified("svgd_small", "small_step_host", "mean_scalars", "fast_loop")),
    pytest.param("sgd", lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4), "small_fused", marks=unverified("svgd_small", "small_step_host", "mean_scalars", "fast_loop")),
    pytest.param("adam", lambda ps: torch.optim.Adam(ps, lr=1e-3), "small_fused", marks=unverified("svgd_small", "small_step_host", "mean_scalars", "fast_loop")),
    # the reference's OWN constructor call, no extra keyword (experiments/iwildcam/models.py:120): fuse_base_optimizer="auto"
    ("adam", lambda ps: torch.optim.Adam(ps, lr=1e-3), "default"),
    ("sgd", lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4), "default"),
    ("adam_wd", lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-2), "default"),
])
def test_svgd_trajectory(golden, backend, name, make_opt, fuse):
    ops, dev = backend
    g = golden(f"svgd_traj_{name}.npz")
    m = int(g["m"])
    model = make_mlp().to(dev)
    params = list(model.parameters())
    init = T(g["init"]).to(dev)
    set_flat(params, init[0])
    k = [0]

    def reset():            # the reference's reset closure produced particle k+1
        k[0] += 1
        set_flat(params, init[k[0]])

    base = make_opt(model.parameters())
    extra = {} if fuse == "default" else dict(
        fuse_base_optimizer=fuse not in (False, "small"), reuse_gram=fuse == "reuse_gram",
        **(SMALL if str(fuse).startswith("small") else {}))
    opt = bde.SVGDOptimizer(model.parameters(), reset, base, particle_count=m, dataset_size=64,
                            l2_reg=float(g["l2_reg"]), kernel_grad_scale=float(g["scale"]), _ops=ops, **extra)
    if fuse == "default":
        assert opt._fuse and not opt._reuse_gram     # plain SGD / Adam over the model's parameters: fused by default
    # a default-constructed optimizer launches device-verified kernels only: the small-model kernel when asked for, or once
    # device_verified.json records a green device run of its sources
    from beyond_deep_ensembles_amd import device_verified
    assert opt._small_model(m, opt._layout.d) == (str(fuse).startswith("small") or device_verified.enabled("svgd_small"))
    fuse = fuse not in (False, "small")
    assert torch.equal(opt.particles.cpu(), init.cpu())
    for i in range(m):      # reference state keys (svgd.py:57)
        assert f"particle_{i}" in opt.state[params[0]]
    x, y = T(g["x"]).to(dev), T(g["y"]).to(dev)
    model64 = make_mlp().double()

    def loss64(particles, xb, yb):
        """Mean particle loss evaluated in fp64 at the given fp32 particles: the anchor for the returned loss."""
        total = 0.0
        for row in particles:
            set_flat(list(model64.parameters()), row.double())
            total += float(F.mse_loss(model64(xb.double().cpu()), yb.double().cpu()))
        return total / len(particles)

    for t in range(g["traj"].shape[0]):
        xb, yb = x[(t % 4) * 16:(t % 4 + 1) * 16], y[(t % 4) * 16:(t % 4 + 1) * 16]
        before = opt.particles.cpu().clone()
        loss = opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())
        # The model forward is stock torch (GPU kernels here, CPU kernels in the reference), so the returned loss is
        # anchored on fp64: our fp32 loss at OUR particles may deviate from the fp64 loss at those particles by at
        # most twice what the reference's fp32 loss deviates from the fp64 loss at ITS particles (floor: 4 ulp).
        ref_before = T(g["traj"][t - 1]) if t > 0 else init.cpu()
        err_ref = abs(g["losses"][t] - loss64(ref_before, xb, yb))
        err = abs(float(loss) - loss64(before, xb, yb))
        assert err <= max(2 * err_ref, 5e-7 * abs(g["losses"][t])), (t, err, err_ref)
        assert abs(float(loss) - g["losses"][t]) <= 2e-5 * abs(g["losses"][t]) + 1e-7     # and stays near the recorded value
        got = opt.particles.cpu().numpy()
        np.testing.assert_allclose(got, g["traj"][t], rtol=2e-5, atol=3e-6)
    if not fuse and float(g["base_step_count"]) >= 0:   # shared optimizer state advanced M times per step (Q5)
        assert float(base.state[params[0]]["step"]) == float(g["base_step_count"])
    # after step() the model aliases the LAST particle (svgd.py:96)
    np.testing.assert_allclose(flat(params).cpu().numpy(), g["model_after"], rtol=2e-5, atol=3e-6)
    # sample_parameters cycles through the particles without copying (svgd.py:107-112)
    for i in range(m + 1):
        opt.sample_parameters()
        assert params[0].data_ptr() == opt.state[params[0]][f"particle_{i % m}"].data_ptr()
    assert opt.get_base_optimizer() is base


@pytest.mark.parametrize("variant", ["default_adam", "unfused_adam", "default_sgd", "reuse_gram_sgd"])
def test_svgd_streaming_path_through_the_shell(backend, variant):
    """SVGDOptimizer.step on a multi-tensor model ABOVE the small-model kernel's limit (D = 669,482 >
    524,288: 2048 -> 300 -> 182 MLP, 4 tensors), 5 particles (the reference's particle_count, iwildcam.yaml:218), three
    steps: the three-launch streaming path (Gram -> statistics -> segmented combine / fused update reading the gradients
    where autograd left them), default constructor (fused) and fuse_base_optimizer=False.  Checked per step against the
    oracle from the SAME particles and gradients: -phi by oracle.svgd_phi, then particle_count shared-state applications
    of a CPU torch optimizer (oracle.svgd_apply_shared_optimizer, svgd.py:92-103), in fp32 (= the reference's
    arithmetic) and in fp64 (the anchor): |ours - fp64| <= max(2 |ref32 - fp64|, 3e-6 max|step|)."""
    if getattr(backend[0], "name", "") == "hip_emu" and variant in ("unfused_adam", "default_sgd"):
        pytest.skip("on the CPU model two of the four variants (10 s each): the same kernels as the other two")
    import oracle.bde_oracle as O
    ops, dev = backend
    torch.manual_seed(21)
    model = nn.Sequential(nn.Linear(2048, 300), nn.Tanh(), nn.Linear(300, 182)).to(dev)
    params = list(model.parameters())
    m, n_data, l2 = 5, 129809.0, 1e-5

    def make_base(ps):
        if variant.endswith("adam"):
            return torch.optim.Adam(ps, lr=1e-3, eps=1e-6, weight_decay=1e-2)
        return torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4)
    base = make_base(params)
    kw = {"unfused_adam": dict(fuse_base_optimizer=False), "reuse_gram_sgd": dict(fuse_base_optimizer=True, reuse_gram=True)}.get(variant, {})
    opt = bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model), base, particle_count=m,
                            dataset_size=n_data, l2_reg=l2, _ops=ops, **kw)
    assert not ops.svgd_small_supported(m, opt._layout.d)          # the streaming path, not the small-model kernel
    assert bool(opt._fuse) == (variant != "unfused_adam")
    shapes = [tuple(p.shape) for p in params]
    numels = [p.numel() for p in params]

    def split(row):
        return [t.view(sh) for t, sh in zip(row.split(numels), shapes)]
    # the oracle's two tracks: CPU model parameters + a CPU base optimizer whose state all particles share
    tracks = {}
    for dt in (torch.float32, torch.float64):
        cpu_params = [torch.nn.Parameter(torch.zeros(sh, dtype=dt)) for sh in shapes]
        tracks[dt] = (cpu_params, make_base(cpu_params))
    x, y = torch.randn(48, 2048, device=dev), torch.randint(0, 182, (48,), device=dev)
    grads = []

    def backward(loss):
        loss.backward()
        grads.append(torch.cat([p.grad.detach().reshape(-1) for p in params]).cpu())
    for t in range(3):
        xb, yb = x[t * 16:(t + 1) * 16], y[t * 16:(t + 1) * 16]
        before = opt.particles.cpu().clone()
        grads.clear()
        loss = opt.step(lambda: F.cross_entropy(model(xb), yb), backward)
        assert torch.isfinite(loss)
        G = torch.stack(grads)
        after = opt.particles.cpu()
        want = {}
        for dt, (cpu_params, cpu_base) in tracks.items():
            P = before.to(dt).clone()
            neg_phi = -O.svgd_phi(P, G.to(dt), l2, 1.0, n_data)
            rows = [split(P[i]) for i in range(m)]
            O.svgd_apply_shared_optimizer(rows, [split(neg_phi[i]) for i in range(m)], cpu_params, cpu_base)
            want[dt] = P
        step_size = float((want[torch.float64] - before.double()).abs().max())
        err_ref = float((want[torch.float32].double() - want[torch.float64]).abs().max())
        err = float((after.double() - want[torch.float64]).abs().max())
        assert step_size > 0
        assert err <= max(2 * err_ref, 3e-6 * step_size), (variant, t, err, err_ref, step_size)
    # the model aliases the last particle afterwards (svgd.py:96) and the shared optimizer advanced M times per step (Q5)
    np.testing.assert_array_equal(flat(params).cpu().numpy(), opt.particles[m - 1].cpu().numpy())
    if variant.endswith("adam"):
        assert float(base.state_dict()["state"][0]["step"]) == 3 * m


# the 96 parameter tensors of the reference's CIFAR model (experiments/cifar/models.py: ResNet20(32, 3, 10, "swish", "frn"),
# 273,610 elements; shapes read off the imported reference, written out here because the reference cannot travel)
@unverified("svgd_small", "small_step_host", "mean_scalars", "fast_loop")
@pytest.mark.parametrize("backend", ["emu", pytest.param("hip", marks=pytest.mark.gpu)], indirect=True)
def test_r5_graph_replay_of_the_small_model_step_changes_nothing(backend, monkeypatch):
    """SVGDOptimizer(graph_replay=True): the small-model step's launches (table upload, gradient packing, the update's two
    launches) recorded once per (staging slot, step scalars) and replayed -- the reference's CIFAR loop
    (experiments/cifar/cifar.py:155-172: SGD, one scheduler step per epoch).  Same kernels, same arguments: the particles
    after every step equal the eager path's bit for bit, through a learning-rate change, and the step really is replayed.
    On the device the recording is a hipGraph; on the CPU model a stand-in that re-issues the recorded launches (the
    bookkeeping around it -- slot rotation, eager first steps, keys, the optimizer's state -- is the product's)."""
    ops, dev = backend
    from beyond_deep_ensembles_amd.svgd import SVGDOptimizer
    if dev.type == "cpu":
        state = {"capturing": None}

        class Recording:
            def __init__(self):
                self.launches = None

            def replay(self):
                self.launches()

        class recording_into:
            def __init__(self, graph, capture_error_mode=None):
                assert capture_error_mode == "thread_local"                  # another thread's HIP calls must not break it
                self.graph = graph

            def __enter__(self):
                if state.get("fail"):
                    raise RuntimeError("operation not permitted when stream is capturing")
                state["capturing"] = self.graph

            def __exit__(self, *exc):
                state["capturing"] = None
        real = SVGDOptimizer._small_sgd_launches

        def launches(self, *a):
            if state["capturing"] is None:
                return real(self, *a)
            state["capturing"].launches = lambda: real(self, *a)           # recorded, not executed (as under a capture)
        monkeypatch.setattr(SVGDOptimizer, "_small_sgd_launches", launches)
        monkeypatch.setattr(SVGDOptimizer, "_graphs_possible", lambda self: True)
        monkeypatch.setattr(torch.cuda, "CUDAGraph", Recording)
        monkeypatch.setattr(torch.cuda, "graph", recording_into)
    torch.manual_seed(5)
    x, y = torch.randn(64, 13, device=dev), torch.randn(64, 1, device=dev)
    init = [torch.randn(13 * 50 + 50 + 50 + 1) * 0.1 for _ in range(4)]

    def run(graph_replay, per_step_schedule=False, steps=14, reload_at=None):
        model = make_mlp().to(dev)
        params = list(model.parameters())
        set_flat(params, init[0].to(dev))
        k = [0]

        def reset():
            k[0] += 1
            set_flat(params, init[k[0]].to(dev))
        base = torch.optim.SGD(model.parameters(), lr=0.004, momentum=0.9, nesterov=True, weight_decay=3e-4)
        opt = bde.SVGDOptimizer(model.parameters(), reset, base, particle_count=4, dataset_size=64, l2_reg=1e-3, _ops=ops,
                                graph_replay=graph_replay, single_launch="two", host_fast_paths=True)
        assert opt._fuse and ops.svgd_small_supported(4, opt._layout.d)
        out, losses = [], []
        for t in range(steps):
            if t == reload_at:
                # an in-process reload mid-run: load_state_dict gives the fused update NEW momentum buffers; a
                # recording made with the old ones must never be replayed (stale addresses: freed memory on the device)
                saved = copy.deepcopy(opt.state_dict())
                old_buf = opt._fused_state["buf"]
                opt.load_state_dict(saved)
                assert not opt._graphs and opt._graph_eager_steps == 0
                keep_alive.append(old_buf)                                   # (so that the allocator cannot hand it out again)
            if t == 9 and not per_step_schedule:                            # the scheduler's epoch step
                for group in base.param_groups:
                    group["lr"] = 0.002
            if per_step_schedule:                                           # a scheduler that steps with every batch
                for group in base.param_groups:
                    group["lr"] = 0.004 * 0.97 ** t
            xb, yb = x[(t % 4) * 16:(t % 4 + 1) * 16], y[(t % 4) * 16:(t % 4 + 1) * 16]
            losses.append(opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward()).item())
            out.append(opt.particles.cpu().clone())
        return out, losses, opt, base
    keep_alive = []
    eager, eager_losses, opt_e, _ = run(False)
    replayed, losses, opt, base = run(True)
    assert opt_e._graph_replays == 0
    assert opt._native_mean_losses() is not None     # the returned mean loss came from host.cpp mean_losses -> bde_mean_scalars
    assert opt_e._small_step_native                  # the eager steps went through host.cpp small_step_sgd
    # three eager steps (the momentum buffers' first one among them), then one recording per staging slot and learning
    # rate: 3 + 3 recordings, every step from the fourth on replayed
    assert opt._graph_replays == 11 and opt._graph_captures == 6, (opt._graph_replays, opt._graph_captures)
    for t, (a, b) in enumerate(zip(eager, replayed)):
        assert torch.isfinite(a).all() and torch.equal(a, b), t
    assert losses == eager_losses
    # a per-STEP scheduler: every step has scalars of its own, a recording never pays -- the option turns itself off after a
    # few recordings (and the results are the eager path's all along)
    eager_s, _, _, _ = run(False, per_step_schedule=True, steps=20)
    replayed_s, _, opt_s, _ = run(True, per_step_schedule=True, steps=20)
    assert 8 <= opt_s._graph_captures <= 10 and opt_s._graph_replays == opt_s._graph_captures
    for t, (a, b) in enumerate(zip(eager_s, replayed_s)):
        assert torch.isfinite(a).all() and torch.equal(a, b), t
    # load_state_dict in the middle of a replayed run: recordings dropped, eager steps again, then new recordings that hold
    # the NEW buffers -- the trajectory is the uninterrupted eager one, and the published momentum buffers keep advancing
    reloaded, losses_r, opt_r, base_r = run(True, reload_at=8)
    for t, (a, b) in enumerate(zip(eager, reloaded)):
        assert torch.isfinite(a).all() and torch.equal(a, b), t
    assert losses_r == eager_losses and opt_r._graph_replays >= 5
    assert all(key[2] == opt_r._fused_state["buf"].data_ptr() for key in opt_r._graphs)
    assert torch.equal(torch.cat([base_r.state[p]["momentum_buffer"].reshape(-1) for p in opt_r._plist]).cpu(),
                       torch.cat([base.state[p]["momentum_buffer"].reshape(-1) for p in opt._plist]).cpu())
    if dev.type == "cpu":
        # a capture the runtime refuses: the option switches itself off with a warning, the step is run eagerly, results unchanged
        state["fail"] = True
        with pytest.warns(UserWarning, match="recording the step failed"):
            failed, losses_f, opt_f, _ = run(True)
        state["fail"] = False
        assert opt_f._graph_replay is False and opt_f._graph_replays == 0
        for t, (a, b) in enumerate(zip(eager, failed)):
            assert torch.equal(a, b), t
    # the base optimizer's published state is the fused buffers', as on the eager path
    assert torch.equal(torch.cat([base.state[p]["momentum_buffer"].reshape(-1) for p in opt._plist]).cpu(),
                       torch.cat([opt_e.state["__base_optimizer"].state[p]["momentum_buffer"].reshape(-1)
                                  for p in opt_e._plist]).cpu())


@pytest.mark.parametrize("path", ["torch_adds", pytest.param("one_launch", marks=unverified("svgd_small", "small_step_host", "mean_scalars", "fast_loop"))])
@pytest.mark.parametrize("loss_dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("m", [3, 8])
def test_r5_returned_loss_is_the_reference_accumulation(backend, loss_dtype, m, path):
    """svgd.py:66,72,105: ``total_loss = tensor(0.0); total_loss += loss`` per particle; ``return total_loss / particle_count``.
    "torch_adds" (the default until bde_mean_scalars is device-verified): torch's in-place adds and division, the
    reference's own ops.  "one_launch": fp32 losses take ONE launch (bde_mean_scalars through host.cpp mean_losses where the
    backend has a C entry point), any other loss type torch's adds.  Both must return exactly what the reference's
    accumulation gives on the same losses ON THE SAME DEVICE (a GPU divides by multiplying with fl(1 / count), and so does
    the kernel since ABI 406; on the CPU model the comparison is against torch's CPU division: equal for m = 8, within one
    ulp for m = 3)."""
    ops, dev = backend
    torch.manual_seed(11)
    model = make_mlp().to(dev)
    x, y = torch.randn(16, 13, device=dev), torch.randn(16, 1, device=dev)
    base = torch.optim.SGD(model.parameters(), lr=1e-3)

    def reset():
        with torch.no_grad():
            for p in model.parameters():
                p.add_(torch.randn_like(p) * 0.05)
    opt = bde.SVGDOptimizer(model.parameters(), reset, base, particle_count=m, dataset_size=64, _ops=ops,
                            **(SMALL if path == "one_launch" else {}))
    seen = []

    def forward():
        loss = F.mse_loss(model(x), y).to(loss_dtype)
        seen.append(loss.detach().clone())
        return loss
    for _ in range(2):
        seen.clear()
        got = opt.step(forward, lambda l: l.backward())
        total = torch.tensor(0.0, device=dev)
        for l in seen:
            total += l
        want = total / m
        assert got.dtype == torch.float32 and len(seen) == m
        kernel_mean = path == "one_launch" and loss_dtype == torch.float32 and hasattr(ops, "mean_scalars")
        if m == 8 or not (kernel_mean and dev.type == "cpu"):
            # torch's own ops on both sides, or the kernel against torch's GPU kernel (both multiply by fl(1 / m))
            assert torch.equal(got.cpu(), want.cpu()), (float(got), float(want))
        else:
            assert abs(float(got) - float(want)) <= 2.4e-7 * abs(float(want))


def _cifar_resnet20_shapes():
    def conv_frn(cout, cin):                                     # a 3x3 convolution, then the FRN layer's four tensors
        return [(cout, cin, 3, 3), (cout,)] + [(1, cout, 1, 1)] * 3
    shapes = [(16, 3, 3, 3), (16,)]                              # stem
    shapes += conv_frn(16, 16) * 6                               # stage 1: three blocks of two convolutions
    shapes += conv_frn(32, 16) + conv_frn(32, 32) + [(32, 16, 1, 1)] + conv_frn(32, 32) * 4      # stage 2 (1x1 shortcut)
    shapes += conv_frn(64, 32) + conv_frn(64, 64) + [(64, 32, 1, 1)] + conv_frn(64, 64) * 4      # stage 3
    return shapes + [(10, 64), (10,)]                            # classifier


@pytest.mark.parametrize("variant", ["default_sgd", "default_adam", "unfused_sgd",
                                     pytest.param("small_sgd", marks=unverified("svgd_small", "small_step_host", "mean_scalars", "fast_loop")), pytest.param("small_adam", marks=unverified("svgd_small", "small_step_host", "mean_scalars", "fast_loop")),
                                     pytest.param("small_unfused_sgd", marks=unverified("svgd_small", "small_step_host", "mean_scalars", "fast_loop"))])
def test_r5_cifar_resnet20_sized_svgd_step_through_the_shell(backend, variant):
    """BASELINE configs[1] at its REAL size through the product shell: the 96 parameter tensors of the reference's CIFAR
    ResNet-20 (273,610 elements), 8 particles, the base optimizer of cifar.yaml (nesterov SGD, momentum 0.9, weight decay) /
    Adam.  "default_*" / "unfused_sgd": what a default-constructed optimizer runs today -- the streaming kernels (Gram ->
    statistics -> fused update reading the gradients where autograd left them / combine + 8 torch steps).  "small_*": gather_seg
    -> the small-model kernel's two launches with the fused shared-state optimizer applications, or its combine form + 8 torch
    steps -- the default once that kernel is device-verified.  Two steps (the first and a later
    application of the shared momentum), each checked against the oracle from the SAME particles and gradients in fp32 and
    fp64: |ours - fp64| <= max(2 |ref32 - fp64|, 3e-6 max|step|) (svgd.py:14-32,82-103)."""
    if getattr(backend[0], "name", "") == "hip_emu" and variant not in ("default_sgd", "small_sgd"):
        pytest.skip("on the CPU model one variant per kernel family (40 s each): the others run the same kernels' other template "
                    "forms, which test_svgd_small_model_fused_step / test_svgd_fused_optimizers_match_torch_shared_state cover")
    import oracle.bde_oracle as O
    ops, dev = backend
    torch.manual_seed(33)
    shapes = _cifar_resnet20_shapes()
    assert len(shapes) == 96 and sum(int(np.prod(sh)) for sh in shapes) == 273_610
    params = [nn.Parameter(torch.randn(sh, device=dev) * 0.05) for sh in shapes]
    m, n_data, l2 = 8, 50000.0, 3e-4

    def reset():
        with torch.no_grad():
            for p in params:
                p.copy_(torch.randn_like(p) * 0.05)

    def make_base(ps):
        if variant.endswith("adam"):
            return torch.optim.Adam(ps, lr=1e-3, weight_decay=5e-4)
        return torch.optim.SGD(ps, lr=0.1, momentum=0.9, nesterov=True, weight_decay=5e-4)
    base = make_base(params)
    kw = dict(fuse_base_optimizer=False) if "unfused" in variant else {}
    if variant.startswith("small"):
        kw.update(SMALL)
    opt = bde.SVGDOptimizer(params, reset, base, particle_count=m, dataset_size=n_data, l2_reg=l2, _ops=ops, **kw)
    assert ops.svgd_small_supported(m, opt._layout.d)              # the small-model kernel's range
    from beyond_deep_ensembles_amd import device_verified
    assert opt._small_model(m, opt._layout.d) == (variant.startswith("small") or device_verified.enabled("svgd_small"))
    assert bool(opt._fuse) == ("unfused" not in variant)
    numels = [p.numel() for p in params]

    def split(row):
        return [t.view(sh) for t, sh in zip(row.split(numels), shapes)]
    tracks = {}
    for dt in (torch.float32, torch.float64):
        cpu_params = [torch.nn.Parameter(torch.zeros(sh, dtype=dt)) for sh in shapes]
        tracks[dt] = (cpu_params, make_base(cpu_params))
    coef = [torch.randn(sh, device=dev) * 0.01 for sh in shapes]
    grads = []

    def forward():                                                  # a loss whose gradient differs per particle: a + 0.3 p
        return sum((a * p).sum() + 0.15 * (p * p).sum() for a, p in zip(coef, params))

    def backward(loss):
        loss.backward()
        grads.append(torch.cat([p.grad.detach().reshape(-1) for p in params]).cpu())
    for t in range(2):
        before = opt.particles.cpu().clone()
        grads.clear()
        loss = opt.step(forward, backward)
        assert torch.isfinite(loss)
        G = torch.stack(grads)
        after = opt.particles.cpu()
        want = {}
        for dt, (cpu_params, cpu_base) in tracks.items():
            P = before.to(dt).clone()
            neg_phi = -O.svgd_phi(P, G.to(dt), l2, 1.0, n_data)
            rows = [split(P[i]) for i in range(m)]
            O.svgd_apply_shared_optimizer(rows, [split(neg_phi[i]) for i in range(m)], cpu_params, cpu_base)
            want[dt] = P
        step_size = float((want[torch.float64] - before.double()).abs().max())
        err_ref = float((want[torch.float32].double() - want[torch.float64]).abs().max())
        err = float((after.double() - want[torch.float64]).abs().max())
        assert step_size > 0
        assert err <= max(2 * err_ref, 3e-6 * step_size), (variant, t, err, err_ref, step_size)
        # the returned loss: mean of the particle losses (svgd.py:105), summed particle by particle in fp32
        want_loss = float(sum(float(sum((a.cpu().double() * p).sum() + 0.15 * (p * p).sum()
                                        for a, p in zip(coef, split(before[i].double())))) for i in range(m)) / m)
        assert abs(float(loss) - want_loss) <= 2e-5 * abs(want_loss) + 1e-6, (float(loss), want_loss)
    np.testing.assert_array_equal(flat(params).cpu().numpy(), opt.particles[m - 1].cpu().numpy())


def test_step_hooks_and_profiler_ranges_still_work(backend):
    """BayesianOptimizer.step skips torch's per-call wrapper (profiler range + hook dispatch: a third of a small model's
    step in host time) only while it has nothing to do: a step pre / post hook registered on the optimizer (or globally)
    fires exactly as on any torch optimizer, and under an active profiler the "Optimizer.step#..." range is recorded."""
    ops, dev = backend
    torch.manual_seed(2)
    model = make_mlp().to(dev)
    base = torch.optim.SGD(model.parameters(), lr=0.05)
    opt = bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model), base, particle_count=3, dataset_size=16,
                            _ops=ops)
    x, y = torch.randn(8, 13, device=dev), torch.randn(8, 1, device=dev)
    step = lambda: opt.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward())
    assert getattr(type(opt).step, "hooked", False)          # torch does not wrap it a second time
    step()
    seen = []
    h1 = opt.register_step_pre_hook(lambda o, args, kwargs: seen.append("pre"))
    h2 = opt.register_step_post_hook(lambda o, args, kwargs: seen.append("post"))
    step()
    assert seen == ["pre", "post"]
    h1.remove()
    h2.remove()
    step()
    assert seen == ["pre", "post"]
    if dev.type == "cpu":
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
            step()
        assert any("Optimizer.step#SVGDOptimizer.step" in e.key for e in prof.key_averages())
    # the combinator has no Optimizer state of its own: its step takes the light path too
    ll = bde.LastLayerBayesianOptimizer(opt, torch.optim.SGD([torch.nn.Parameter(torch.zeros(2, device=dev))], lr=0.1))
    assert torch.isfinite(ll.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward()))


def test_svgd_many_particles(backend):
    """particle_count > 16: the blocked update kernel, then (fuse_base_optimizer) ONE launch that applies the base optimizer
    to all particles in order with its shared state -- the same trajectory as the reference's loop of base.step() calls."""
    ops, dev = backend
    torch.manual_seed(3)
    model = nn.Linear(7, 2).to(dev)
    base = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    opt = bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model), base, particle_count=20,
                            dataset_size=50, l2_reg=0.01, fuse_base_optimizer=True, _ops=ops)
    x, y = torch.randn(10, 7, device=dev), torch.randn(10, 2, device=dev)
    before = opt.particles.clone()
    for _ in range(2):
        loss = opt.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward())
    assert torch.isfinite(loss) and torch.isfinite(opt.particles).all()
    assert not torch.equal(before, opt.particles)
    # same two steps with the per-tensor reference arithmetic (oracle) on the CPU
    import oracle.bde_oracle as O
    P = before.cpu().clone()
    buf = None
    xc, yc = x.cpu(), y.cpu()
    for _ in range(2):
        G = torch.zeros_like(P)
        for i in range(20):
            w, b = P[i, :14].view(2, 7).clone().requires_grad_(), P[i, 14:].clone().requires_grad_()
            F.mse_loss(F.linear(xc, w, b), yc).backward()
            G[i] = torch.cat([w.grad.flatten(), b.grad])
        neg_phi = -O.svgd_phi(P, G, 0.01, 1.0, 50.0)
        for i in range(20):
            g = neg_phi[i]
            buf = g.clone() if buf is None else 0.9 * buf + g
            P[i] -= 0.1 * buf
    np.testing.assert_allclose(opt.particles.cpu().numpy(), P.numpy(), rtol=2e-4, atol=2e-6)


def test_svgd_state_dict_roundtrip(backend):
    ops, dev = backend
    torch.manual_seed(0)
    model = make_mlp().to(dev)
    mk = lambda: bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model),
                                   torch.optim.SGD(model.parameters(), lr=0.1), particle_count=3, dataset_size=10,
                                   _ops=ops)
    a = mk()
    want = a.particles.clone()
    sd = a.state_dict()
    b = mk()
    assert not torch.equal(b.particles, want)
    b.load_state_dict(sd)
    assert torch.equal(b.particles, want)
    assert b.state[list(model.parameters())[0]]["particle_1"].data_ptr() == b._pviews[1][0].data_ptr()


@pytest.mark.parametrize("algo,base_kind", [("svgd", "sgd"), ("svgd", "adam"), ("svgd_unfused", "sgd"), ("swag", "sgd"), ("swag", "adam")])
def test_resuming_from_a_pickled_checkpoint_continues_the_run(backend, algo, base_kind, tmp_path):
    """torch.save(model / optimizer state_dict) -> a NEW model + base optimizer + shell -> torch.load -> load_state_dict, as
    the reference's drivers resume (iwildcam.py:84-88, ensemble.py:23-26): the resumed run takes the same steps as the
    uninterrupted one.  The state carries the checkpoint's pickled base optimizer (svgd.py:51, swag.py:28), bound to ITS copies
    of the parameters; the shell keeps the live base optimizer and hands it the loaded state (momentum / Adam moments, step
    counts) -- without that a resumed run would step tensors nobody looks at."""
    ops, dev = backend
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(64, 13, generator=g).to(dev), torch.randn(64, 1, generator=g).to(dev)

    def make(seed):
        torch.manual_seed(seed)
        model = make_mlp().to(dev)
        base = (torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4) if base_kind == "sgd"
                else torch.optim.Adam(model.parameters(), lr=1e-2))
        if algo.startswith("svgd"):
            opt = bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model), base, particle_count=3,
                                    dataset_size=64, l2_reg=0.01, fuse_base_optimizer="auto" if algo == "svgd" else False, _ops=ops)
        else:
            opt = bde.SwagOptimizer(model.parameters(), base, update_interval=1, start_epoch=0, deviation_samples=4, _ops=ops)
        return model, opt

    def run(model, opt, steps):
        out = []
        for t in steps:
            xb, yb = x[(t % 4) * 16:(t % 4 + 1) * 16], y[(t % 4) * 16:(t % 4 + 1) * 16]
            out.append(float(opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward()).detach()))
        return out
    m1, o1 = make(1)
    run(m1, o1, range(3))
    path = str(tmp_path / "ckpt.pt")
    torch.save({"model": m1.state_dict(), "optimizer": o1.state_dict()}, path)
    want = run(m1, o1, range(3, 6))                                   # the uninterrupted run goes on
    m2, o2 = make(2)                                                  # a new process would build these from scratch
    live_base = o2.get_base_optimizer()
    ck = torch.load(path, weights_only=False)
    m2.load_state_dict(ck["model"])
    o2.load_state_dict(ck["optimizer"])
    assert o2.get_base_optimizer() is live_base                       # still the optimizer over THIS model's parameters
    got = run(m2, o2, range(3, 6))
    np.testing.assert_allclose(got, want, rtol=2e-5)
    if algo.startswith("svgd"):
        np.testing.assert_allclose(o2.particles.cpu().numpy(), o1.particles.cpu().numpy(), rtol=2e-5, atol=2e-6)
    else:
        for a, b in zip(m1.parameters(), m2.parameters()):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(o2.state["__mean"].cpu().numpy(), o1.state["__mean"].cpu().numpy(), rtol=2e-5, atol=2e-6)


# ------------------------------------------------------------------ SWAG --
def test_swag_schedule_bit_exact(golden, backend):
    ops, dev = backend
    g = golden("swag_schedule.npz")
    for ci, (steps_per_epoch, start_epoch, interval, epochs) in enumerate(g["cfgs"]):
        p = nn.Parameter(torch.zeros(3, device=dev))
        base = torch.optim.SGD([p], lr=1.0)
        opt = bde.SwagOptimizer([p], base, update_interval=float(interval), start_epoch=int(start_epoch),
                                deviation_samples=4, _ops=ops)
        trace = []
        for e in range(int(epochs)):
            for b in range(int(steps_per_epoch)):
                opt.step(lambda: p.sum(), lambda l: l.backward())
                trace.append([e, b, opt.state["__epoch"], opt.state["__steps_since_swag_start"], opt.state["__updates"]])
            opt.complete_epoch()
        np.testing.assert_array_equal(np.array(trace, dtype=np.int64), g[f"trace_{ci}"])


def test_swag_statistics_columns_and_samples(golden, backend):
    ops, dev = backend
    g = golden("swag_stats.npz")
    for ci, (total_updates, tagged, lr, interval, k) in enumerate(g["cases"]):
        k = int(k)
        theta0 = T(g[f"theta0_{ci}"])
        c = T(g[f"c_{ci}"]).to(dev)
        p1 = nn.Parameter(theta0[:7].clone().to(dev))
        p2 = nn.Parameter(theta0[7:].clone().view(2, 3).to(dev))
        base = torch.optim.SGD([p1, p2], lr=float(lr))
        opt = bde.SwagOptimizer([p1, p2], base, update_interval=int(interval), start_epoch=0, deviation_samples=k,
                                _ops=ops)
        c1, c2 = c[:7], c[7:].view(2, 3)
        fwd = lambda: (p1 * c1).sum() + (p2 * c2).sum()
        for t in range(int(total_updates) * int(interval)):
            opt.step(fwd, lambda l: l.backward())
            np.testing.assert_array_equal(flat([p1, p2]).cpu().numpy(), g[f"thetas_{ci}"][t])
        assert opt.state["__updates"] == int(total_updates)
        # statistics in the reference's layout: bit-exact (iterate <-> column mapping included)
        np.testing.assert_array_equal(opt.mean_vector().cpu().numpy(), g[f"mean_{ci}"])
        np.testing.assert_array_equal(opt.sq_vector().cpu().numpy(), g[f"sq_{ci}"])
        np.testing.assert_array_equal(opt.deviations_dk().cpu().numpy(), g[f"dev_{ci}"])
        # samples with the recorded noise (eps_W first, then eps_D)
        before = flat([p1, p2]).clone()
        for s in range(3):
            ew, ed = T(g[f"eps_w_{ci}"][s]).to(dev), T(g[f"eps_d_{ci}"][s]).to(dev)
            opt.noise_source = lambda kk, dd: (ew, ed)
            opt.sample_parameters()
            want = g[f"samples_{ci}"][s]
            mag = np.abs(g[f"mean_{ci}"]) + np.abs(want) + 1e-3
            assert np.max(np.abs(flat([p1, p2]).cpu().numpy() - want) / mag) < 2e-6
            assert opt.state["__params_dirty"]
        # the next step() first restores the training weights (swag.py:38)
        opt.step(fwd, lambda l: l.backward())
        np.testing.assert_array_equal(flat([p1, p2]).cpu().numpy(), g[f"theta_after_restore_step_{ci}"])
        assert not opt.state["__params_dirty"]
        # checkpoint in the reference's wire layout, and back
        sd = opt.state_dict()
        assert tuple(sd["state"]["__deviations"].shape) == (13, k)
        np.testing.assert_array_equal(sd["state"]["__deviations"].numpy(), opt.deviations_dk().cpu().numpy())
        q1, q2 = nn.Parameter(p1.detach().clone()), nn.Parameter(p2.detach().clone())
        opt2 = bde.SwagOptimizer([q1, q2], torch.optim.SGD([q1, q2], lr=float(lr)), update_interval=int(interval),
                                 start_epoch=0, deviation_samples=k, _ops=ops)
        sd2 = {"state": dict(sd["state"]), "param_groups": sd["param_groups"]}
        sd2["state"]["__base_optimizer"] = opt2.state["__base_optimizer"]
        opt2.load_state_dict(sd2)
        np.testing.assert_array_equal(opt2.deviations_dk().cpu().numpy(), opt.deviations_dk().cpu().numpy())
        np.testing.assert_array_equal(opt2.mean_vector().cpu().numpy(), opt.mean_vector().cpu().numpy())
        assert opt2.state["__updates"] == opt.state["__updates"]


def test_swag_torch_rng_matches_reference_stream(backend):
    """rng='torch' consumes the generator exactly as LowRankMultivariateNormal.rsample does."""
    ops, dev = backend
    torch.manual_seed(0)
    p = nn.Parameter(torch.randn(37, device=dev))
    opt = bde.SwagOptimizer([p], torch.optim.SGD([p], lr=0.1), update_interval=1, deviation_samples=3, _ops=ops)
    for _ in range(4):
        opt.step(lambda: (p ** 2).sum(), lambda l: l.backward())
    torch.manual_seed(123)
    opt.sample_parameters()
    got = p.detach().clone()
    torch.manual_seed(123)
    ew = torch.empty(3, device=dev).normal_()
    ed = torch.empty(37, device=dev).normal_()
    opt.noise_source = lambda k, d: (ew, ed)
    opt.sample_parameters()
    assert torch.equal(got, p.detach())


def test_swag_prefetch_equals_one_by_one(backend):
    """DeepEnsemble.predict prefetches a SWAG member's samples in one batched pass (rng="philox");
    the predictions equal those of one-by-one sampling."""
    ops, dev = backend

    def member():
        torch.manual_seed(2)
        model = nn.Linear(6, 3).to(dev)
        opt = bde.SwagOptimizer(model.parameters(), torch.optim.SGD(model.parameters(), lr=0.1), update_interval=1,
                                deviation_samples=4, rng="philox", seed=7, _ops=ops)
        x = torch.randn(8, 6, generator=torch.Generator().manual_seed(1)).to(dev)
        for _ in range(6):
            opt.step(lambda: model(x).pow(2).mean(), lambda l: l.backward())
        return model, opt, x
    m1, o1, x = member()
    m2, o2, _ = member()
    one_by_one = []
    for _ in range(7):
        o1.sample_parameters()
        one_by_one.append(m1(x).detach().clone())
    ens = bde.DeepEnsemble([(m2, o2)])
    batched = ens.predict(lambda m: m(x).detach().clone(), 7)
    assert torch.allclose(torch.stack(one_by_one), batched, rtol=1e-6, atol=1e-7)
    assert o2._sample_counter == 7 and o2._prefetched is None
    # small models keep prefetched samples as contiguous rows, served by a device copy into the sample vector or by
    # re-pointing the parameters at the row (whichever the cost model picks): same predictions either way
    m4, o4, _ = member()
    o4._copy_is_cheaper = not o2._copy_is_cheaper
    assert torch.equal(bde.DeepEnsemble([(m4, o4)]).predict(lambda m: m(x).detach().clone(), 7), batched)
    # large models store them in pieces, like the statistics, and serve them by one streaming copy
    m5, o5, _ = member()
    o5._PIECES_FROM = 0
    assert torch.equal(bde.DeepEnsemble([(m5, o5)]).predict(lambda m: m(x).detach().clone(), 7), batched)
    # sample_batch() hands out the same samples as contiguous rows
    m3, o3, _ = member()
    rows = o3.sample_batch(7)
    served = []
    for s in range(7):
        o3.use_sample(rows[s])
        served.append(m3(x).detach().clone())
    assert torch.allclose(torch.stack(served), batched, rtol=1e-6, atol=1e-7)
    # a training step drops any prefetched rows and restores the training weights
    o2.prefetch_samples(5)
    o2.sample_parameters()
    o2.step(lambda: m2(x).pow(2).mean(), lambda l: l.backward())
    assert o2._prefetched is None and not o2.state["__params_dirty"]


def test_r6_cifar_resnet20_swag_k20_thirty_samples_through_the_shell(backend):
    """BASELINE configs[2] at its real size through the product shell: the 96 parameter tensors of the reference's CIFAR
    ResNet-20 (273,610 elements), SwagOptimizer with K = 20 deviation columns (22 moment updates: the ring has wrapped),
    then DeepEnsemble.predict with 30 posterior samples -- one batched sampling pass (rng="philox") served sample by sample.
    Every sample is checked against the ORACLE's swag.py:57,107-114 restatement fed with the same Philox noise (drawn by the
    checker's Philox, tests/oracle_ops.py), from the statistics the optimizer itself accumulated (which must equal the
    oracle's moment recursion bit for bit, swag.py:98-104); the 30 samples are pairwise different and the training weights
    come back afterwards."""
    import oracle.bde_oracle as O
    ops, dev = backend
    torch.manual_seed(40)
    shapes = _cifar_resnet20_shapes()
    params = [nn.Parameter(torch.randn(sh, device=dev) * 0.05) for sh in shapes]
    d, k, s_count = sum(p.numel() for p in params), 20, 30
    assert d == 273_610
    base = torch.optim.SGD(params, lr=0.05, momentum=0.9)
    opt = bde.SwagOptimizer(params, base, update_interval=1, start_epoch=0, deviation_samples=k, rng="philox", seed=11, _ops=ops)
    st = O.swag_init(flat(params).cpu(), k)
    coef = [torch.randn(sh, device=dev) * 0.01 for sh in shapes]
    for n in range(1, 23):
        opt.step(lambda: sum((c * p).sum() + 0.05 * (p * p).sum() for c, p in zip(coef, params)), lambda l: l.backward())
        st.updates = n
        O.swag_moment_update(st, flat(params).cpu())
    assert torch.equal(opt.mean_vector().cpu(), st.mean) and torch.equal(opt.sq_vector().cpu(), st.sq_weights)
    assert torch.equal(opt.deviations_dk().cpu(), st.deviations)
    trained = flat(params).cpu().clone()
    model = nn.Module()
    model.p = nn.ParameterList(params)
    seen = bde.DeepEnsemble([(model, opt)]).predict(lambda m: flat(m.p).cpu().clone(), s_count)
    assert seen.shape == (s_count, d)
    from tests.oracle_ops import _philox
    from oracle import philox as PH
    for s in range(s_count):
        # the noise of posterior sample s from the numpy Philox checker: one stream per sample (stream id = the optimizer's
        # sample counter), eps_W from the low-rank domain, eps_D from the diagonal one -- swag.py:57's draw order
        eps_w = _philox(opt.seed, s, k, PH.DOMAIN_LOWRANK, PH.SWAG_ROUNDS)         # (the samplers' own round count: 7)
        eps_d = _philox(opt.seed, s, d, rounds=PH.SWAG_ROUNDS)
        want = O.swag_sample(st.mean, st.sq_weights, st.deviations, eps_w, eps_d)
        assert torch.allclose(seen[s], want, rtol=3e-5, atol=3e-6), (s, float((seen[s] - want).abs().max()))
    # samples are draws around the SWAG mean with the posterior's spread, all different
    spread = (seen - st.mean).std(dim=0).mean().item()
    diag = (0.5 * (torch.relu(st.sq_weights - st.mean ** 2) + 1e-6)).sqrt().mean().item()
    assert 0.5 * diag < spread < 4.0 * diag + st.deviations.abs().mean().item(), (spread, diag)
    assert len({float(seen[s, 0]) for s in range(s_count)}) == s_count
    opt.step(lambda: sum((p * p).sum() for p in params) * 0.0, lambda l: l.backward())      # the next step restores the weights
    assert not opt.state["__params_dirty"]


# ------------------------------------------------------------------- BBB --
class LocalReparamLinear(nn.Module):
    """Test model layer: the local-reparameterisation forward of the reference's
    BBBLinear (bbb_layers.py:70-80) on bde.GaussianParameter.  The layer is model
    code (out of scope, stays PyTorch); the noise is replayed from the fixture."""

    def __init__(self, i, o, tape, ops):
        super().__init__()
        self.weight = bde.GaussianParameter((o, i), _ops=ops)
        self.bias = bde.GaussianParameter((o,), _ops=ops)
        self.tape = tape

    def forward(self, x):
        mean = F.linear(x, self.weight.mean, self.bias.mean)
        var = F.linear((x ** 2).clamp(min=1e-4), (self.weight.std ** 2).clamp(min=1e-4),
                       (self.bias.std ** 2).clamp(min=1e-4))
        return mean + torch.sqrt(var) * self.tape.pop(0).to(x.device)


class SampledLinear(nn.Module):
    def __init__(self, i, o, tape, ops):
        super().__init__()
        self.weight = bde.GaussianParameter((o, i), _ops=ops)
        self.bias = bde.GaussianParameter((o,), _ops=ops)
        self.weight.noise_source = lambda rho: tape.pop(0).to(rho.device)
        self.bias.noise_source = lambda rho: tape.pop(0).to(rho.device)

    def forward(self, x):
        return F.linear(x, self.weight.sample(), self.bias.sample())


@pytest.mark.parametrize("tag,layer", [("b", LocalReparamLinear), ("c", SampledLinear)])
def test_bbb_trajectory(golden, backend, tag, layer):
    """BASELINE config #1: UCI-housing-shaped 13-50-1 mean-field MLP, BBBOptimizer with Adam."""
    ops, dev = backend
    g = golden("bbb.npz")
    tape = [T(g[f"{tag}_eps_{i}"]) for i in range(int(g[f"{tag}_n_eps"]))]
    model = nn.Sequential(layer(13, 50, tape, ops), nn.ReLU(), layer(50, 1, tape, ops)).to(dev)
    extra = nn.Parameter(torch.zeros(4, device=dev))
    names = [str(n) for n in g[f"{tag}_names"]]
    named = dict(model.named_parameters())
    named["extra"] = extra
    with torch.no_grad():
        for n in names:
            named[n].copy_(T(g[f"{tag}_init/{n}"]).to(dev))
    params = [named[n] for n in names]
    prior = bde.GaussianPrior(0, 1.0)
    base = torch.optim.Adam(params, lr=1e-2)
    opt = bde.BBBOptimizer(params, base, prior, dataset_size=48, mc_samples=2, kl_rescaling=0.5, components=1,
                           l2_scale=0.3, _ops=ops)
    x, y = T(g[f"{tag}_x"]).to(dev), T(g[f"{tag}_y"]).to(dev)
    for t in range(3):
        xb, yb = x[(t % 3) * 16:(t % 3 + 1) * 16], y[(t % 3) * 16:(t % 3 + 1) * 16]
        loss = opt.step(lambda: F.mse_loss(model(xb), yb) + extra.sum() * 0.01, lambda l: l.backward())
        want = g[f"{tag}_losses"][t]
        assert abs(float(loss) - want) <= 5e-6 * abs(want), (t, float(loss), want)
        np.testing.assert_allclose(flat(params).cpu().numpy(), g[f"{tag}_traj"][t], rtol=1e-4, atol=2e-5)
    assert not tape
    assert opt.get_base_optimizer() is base


def test_bbb_layers_reproduce_reference_bbblinear_trajectory(golden, backend, monkeypatch):
    """bde.BBBLinear (local reparameterisation) + BBBOptimizer against the trajectory the REFERENCE's own
    BBBLinear + BBBOptimizer produced on the UCI-shaped MLP (BASELINE config #1), noise replayed."""
    ops, dev = backend
    import beyond_deep_ensembles_amd.bbb_layers as L
    g = golden("bbb.npz")
    tape = [T(g[f"b_eps_{i}"]) for i in range(int(g["b_n_eps"]))]
    real_normal_like = L.normal_like
    monkeypatch.setattr(L, "normal_like", lambda t: tape.pop(0).to(t.device))
    prior = bde.GaussianPrior(0, 1.0)
    model = nn.Sequential(bde.BBBLinear(13, 50, prior, prior, _ops=ops), nn.ReLU(),
                          bde.BBBLinear(50, 1, prior, prior, _ops=ops)).to(dev)
    extra = nn.Parameter(torch.zeros(4, device=dev))
    names = [str(n) for n in g["b_names"]]
    named = dict(model.named_parameters())
    named["extra"] = extra
    with torch.no_grad():
        for n in names:
            named[n].copy_(T(g[f"b_init/{n}"]).to(dev))
    params = [named[n] for n in names]
    opt = bde.BBBOptimizer(params, torch.optim.Adam(params, lr=1e-2), prior, dataset_size=48, mc_samples=2,
                           kl_rescaling=0.5, components=1, l2_scale=0.3, _ops=ops)
    x, y = T(g["b_x"]).to(dev), T(g["b_y"]).to(dev)
    for t in range(3):
        xb, yb = x[(t % 3) * 16:(t % 3 + 1) * 16], y[(t % 3) * 16:(t % 3 + 1) * 16]
        loss = opt.step(lambda: F.mse_loss(model(xb), yb) + extra.sum() * 0.01, lambda l: l.backward())
        assert abs(float(loss.detach()) - g["b_losses"][t]) <= 5e-6 * abs(g["b_losses"][t])
        np.testing.assert_allclose(flat(params).cpu().numpy(), g["b_traj"][t], rtol=1e-4, atol=2e-5)
    assert not tape
    # lazy layer KL == closed form, eval mode shares one noise draw across the batch, conv layer runs
    model.train()
    want = O_kl(model[0])
    assert abs(float(model[0].kl) - want) <= 1e-5 * abs(want)
    model.eval()
    assert model[0].kl == 0
    monkeypatch.setattr(L, "normal_like", real_normal_like)
    conv = bde.BBBConv2d(3, 4, 3, prior, prior, padding=1, _ops=ops).to(dev).eval()
    out = conv(torch.ones(2, 3, 5, 5, device=dev))
    assert out.shape == (2, 4, 5, 5) and torch.equal(out[0], out[1])       # frozen noise: same sample for the batch
    net = nn.Sequential(nn.Linear(4, 3), nn.ReLU(), nn.Conv2d(1, 2, 3, padding=1)).to(dev)
    assert bde.make_module_bbb(net, prior, _ops=ops) == 2 and isinstance(net[0], bde.BBBLinear)


def O_kl(layer):
    import oracle.bde_oracle as O
    return float(O.gauss_kl(layer.weight.mean.detach().cpu(), layer.weight.rho.detach().cpu(), 0.0, 1.0)
                 + O.gauss_kl(layer.bias.mean.detach().cpu(), layer.bias.rho.detach().cpu(), 0.0, 1.0))


def test_bbb_nan_loss_skips_update(backend):
    ops, dev = backend
    gp = bde.GaussianParameter((5,), _ops=ops).to(dev)
    gp.blundell_init()
    params = list(gp.parameters())
    opt = bde.BBBOptimizer(params, torch.optim.SGD(params, lr=0.1), bde.GaussianPrior(0, 1.0), dataset_size=10, _ops=ops)
    before = gp.mean.detach().clone()
    loss = opt.step(lambda: gp.mean.sum() * float("nan"), lambda l: l.backward())
    assert torch.isnan(loss)
    assert torch.equal(before, gp.mean.detach())      # bbb.py:81: update skipped, loss still returned


# ------------------------------------------------------------------ iVON --
def test_ivon_trajectory(golden, backend):
    ops, dev = backend
    g = golden("ivon.npz")
    for ci, (aug, mc, damping, temp) in enumerate(g["cases"]):
        mc = int(mc)
        model = make_mlp().to(dev)
        params = list(model.parameters())
        set_flat(params, T(g[f"init_{ci}"]).to(dev))
        opt = bde.iVONOptimizer(params, lr=1e-2, prior_prec=50.0, dataset_size=48, damping=float(damping),
                                tempering=float(temp), augmentation=float(aug), mc_samples=mc, _ops=ops)
        eps = [T(e).to(dev) for e in g[f"eps_{ci}"]]
        opt.noise_source = lambda d: eps.pop(0)
        x, y = T(g[f"x_{ci}"]).to(dev), T(g[f"y_{ci}"]).to(dev)
        assert opt.get_base_optimizer() is opt
        for t in range(3):
            xb, yb = x[(t % 3) * 16:(t % 3 + 1) * 16], y[(t % 3) * 16:(t % 3 + 1) * 16]
            loss = opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())
            assert abs(float(loss) - g[f"losses_{ci}"][t]) <= 3e-6 * abs(g[f"losses_{ci}"][t])
            st = lambda key: flat([opt.state[p][key] for p in params]).cpu().numpy()
            np.testing.assert_allclose(st("mean"), g[f"means_{ci}"][t], rtol=3e-5, atol=1e-6)
            np.testing.assert_allclose(st("momentum"), g[f"moms_{ci}"][t], rtol=3e-4, atol=1e-6)
            np.testing.assert_allclose(st("precision"), g[f"precs_{ci}"][t], rtol=3e-5, atol=1e-7)
            # the parameters stay at the last noisy sample (Q13)
            np.testing.assert_allclose(flat(params).cpu().numpy(), g[f"after_{ci}"][t], rtol=3e-5, atol=2e-6)
        opt.sample_parameters()
        np.testing.assert_allclose(flat(params).cpu().numpy(), g[f"eval_sample_{ci}"], rtol=3e-5, atol=2e-6)
        assert not eps


def test_ivon_state_dict_roundtrip(backend):
    ops, dev = backend
    torch.manual_seed(0)
    x, y = torch.randn(16, 13, device=dev), torch.randn(16, 1, device=dev)

    def make():
        model = make_mlp().to(dev)
        opt = bde.iVONOptimizer(model.parameters(), lr=1e-2, prior_prec=50.0, dataset_size=16, mc_samples=2,
                                rng="torch", _ops=ops)
        return model, opt
    m1, o1 = make()
    for _ in range(3):
        o1.step(lambda: F.mse_loss(m1(x), y), lambda l: l.backward())
    sd_model, sd_opt = m1.state_dict(), o1.state_dict()
    m2, o2 = make()
    m2.load_state_dict(sd_model)
    o2.load_state_dict(sd_opt)
    p1, p2 = list(m1.parameters()), list(m2.parameters())
    for a, b in zip(p1, p2):
        for key in ("mean", "momentum", "precision"):
            assert torch.equal(o1.state[a][key], o2.state[b][key])
    assert o2.state[p2[0]]["mean"].data_ptr() == o2._groups[0].mean.data_ptr()      # re-aliased to the flat buffer
    assert o2.param_groups[0]["step"] == 3
    # both continue identically from here
    for o, m in ((o1, m1), (o2, m2)):
        o.noise_source = lambda d, _g=torch.Generator(device=dev).manual_seed(5): torch.randn(d, device=dev, generator=_g)
    l1 = o1.step(lambda: F.mse_loss(m1(x), y), lambda l: l.backward())
    l2 = o2.step(lambda: F.mse_loss(m2(x), y), lambda l: l.backward())
    assert torch.equal(l1, l2)
    for a, b in zip(p1, p2):
        assert torch.equal(o1.state[a]["mean"], o2.state[b]["mean"])


def test_bbb_state_dict_roundtrip(backend, monkeypatch, conv_kw):
    """BBBOptimizer.state_dict() / load_state_dict() (stock Optimizer pickling, as the reference's checkpoints, cifar.py:175-176):
    a second model + optimizer restored from the first continues with the same losses and weights; loading also drops the
    layers' cached sigma^2 / prepared convolution weights (they belong to the weights that were just replaced)."""
    ops, dev = backend
    import beyond_deep_ensembles_amd.bbb_layers as L
    torch.manual_seed(0)
    x, y = torch.randn(16, 1, 8, 8, device=dev), torch.randn(16, 3, device=dev)
    prior = bde.GaussianPrior(0, 1.0)

    def make():
        model = nn.Sequential(bde.BBBConv2d(1, 4, 3, prior, prior, padding=1, _ops=ops, **conv_kw), nn.ReLU(), nn.Flatten(),
                              bde.BBBLinear(256, 3, prior, prior, _ops=ops)).to(dev)
        base = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
        return model, bde.BBBOptimizer(model.parameters(), base, prior, dataset_size=16, mc_samples=2, _ops=ops)
    tape = [torch.randn(16, 4, 8, 8), torch.randn(16, 3)] * 64
    pos = [0]

    def replay(t):
        pos[0] += 1
        return tape[pos[0] - 1][: t.shape[0]].to(t.device).reshape(t.shape)
    monkeypatch.setattr(L, "normal_like", replay)
    m1, o1 = make()
    for _ in range(2):
        o1.step(lambda: F.mse_loss(m1(x), y), lambda l: l.backward())
    sd_model, sd_opt = m1.state_dict(), o1.state_dict()
    m2, o2 = make()
    m2(x)                                                             # fills the new layers' caches with ITS initial weights
    m2.load_state_dict(sd_model)
    epoch = L._SigmaCache.epoch
    o2.load_state_dict(sd_opt)
    assert L._SigmaCache.epoch > epoch                                # load_state_dict invalidates the per-version caches
    start = pos[0]
    l1 = o1.step(lambda: F.mse_loss(m1(x), y), lambda l: l.backward())
    pos[0] = start
    l2 = o2.step(lambda: F.mse_loss(m2(x), y), lambda l: l.backward())
    torch.testing.assert_close(l2, l1, rtol=1e-6, atol=1e-7)
    for a, b in zip(m1.parameters(), m2.parameters()):
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-6)


# -------------------------------------------------------------- ensemble --
class _Counting:
    def __init__(self):
        self.n = 0

    def sample_parameters(self):
        self.n += 1


def test_ensemble_split_and_fan_out(golden):
    from beyond_deep_ensembles_amd.ensemble import fan_out, split_samples
    for r in golden("ensemble.npz")["rows"]:
        samples, members, n_out = int(r[0]), int(r[1]), int(r[2])
        counts = [int(c) for c in r[3:3 + members]]
        pairs = [(nn.Linear(1, 1), _Counting()) for _ in range(members)]
        ens = bde.DeepEnsemble(pairs)
        out = ens.predict(lambda m: torch.zeros(1), samples)
        assert out.shape[0] == n_out and [o.n for _, o in pairs] == counts
        assert split_samples(samples, members) == counts
        for world in (2, 8):
            units = [u for rk in range(world) for u in fan_out(samples, members, rk, world)]
            assert sorted(u[0] for u in units) == list(range(n_out))          # every unit exactly once
            sizes = [len(fan_out(samples, members, rk, world)) for rk in range(world)]
            assert max(sizes) - min(sizes) <= 1                               # balanced
    sd = bde.DeepEnsemble([(nn.Linear(2, 2), torch.optim.SGD(nn.Linear(2, 2).parameters(), lr=0.1))]).state_dict()
    assert set(sd) == {"models", "optimizers"}                                 # ensemble.py:17-21


def test_product_has_no_cpu_path():
    """Without _ops the shells use libbde_hip.so and refuse CPU parameters."""
    p = nn.Parameter(torch.zeros(4))
    with pytest.raises(RuntimeError):
        bde.SwagOptimizer([p], torch.optim.SGD([p], lr=0.1), update_interval=1)


# ------------------------------------------------- GradScaler / combinators --
def _scaler(dev):
    return torch.amp.GradScaler(dev.type, init_scale=1024.0)


@pytest.mark.parametrize("algo", ["svgd", "swag", "bbb", "ivon"])
def test_grad_scaler_path_matches_unscaled(backend, algo):
    """step(..., grad_scaler=scaler) with backward_closure = scaler.scale(loss).backward() (Readme.md:51-55)
    lands on the same parameters as the plain path (power-of-two scale, no overflow)."""
    ops, dev = backend
    x = torch.randn(16, 13, generator=torch.Generator().manual_seed(3)).to(dev)
    y = torch.randn(16, 1, generator=torch.Generator().manual_seed(4)).to(dev)

    def build():
        torch.manual_seed(11)
        if algo == "bbb":
            tape = [torch.randn(16, 50), torch.randn(16, 1)] * 8
            model = nn.Sequential(LocalReparamLinear(13, 50, tape, ops), nn.ReLU(), LocalReparamLinear(50, 1, tape, ops)).to(dev)
            for m in model:
                if hasattr(m, "weight"):
                    m.weight.blundell_init(); m.bias.blundell_init()
        else:
            model = make_mlp().to(dev)
        params = list(model.parameters())
        if algo == "svgd":
            opt = bde.SVGDOptimizer(params, lambda: None, torch.optim.SGD(params, lr=0.05, momentum=0.9),
                                    particle_count=3, dataset_size=16, l2_reg=0.01, _ops=ops)
            with torch.no_grad():       # distinct particles without touching the RNG
                moved = opt.particles.clone()
                moved[1] += 0.01
                moved[2] -= 0.02
                opt.set_particles(moved)
                assert not torch.equal(opt.particles[0], opt.particles[1])
        elif algo == "swag":
            opt = bde.SwagOptimizer(params, torch.optim.SGD(params, lr=0.05), update_interval=1, deviation_samples=3, _ops=ops)
        elif algo == "bbb":
            opt = bde.BBBOptimizer(params, torch.optim.SGD(params, lr=0.05), bde.GaussianPrior(0, 1.0), dataset_size=16,
                                   mc_samples=2, kl_rescaling=0.5, _ops=ops)
        else:
            opt = bde.iVONOptimizer(params, lr=1e-2, prior_prec=50.0, dataset_size=16, mc_samples=2, _ops=ops)
            gen = torch.Generator().manual_seed(5)
            opt.noise_source = lambda d: torch.randn(d, generator=gen).to(dev)
        return model, opt

    results = []
    for use_scaler in (False, True):
        model, opt = build()
        scaler = _scaler(dev) if use_scaler else None
        if scaler is not None:
            opt.init_grad_scaler(scaler)
        for _ in range(3):
            if scaler is None:
                loss = opt.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward())
            else:
                loss = opt.step(lambda: F.mse_loss(model(x), y), lambda l: scaler.scale(l).backward(), grad_scaler=scaler)
                scaler.update()
            assert loss is not None and torch.isfinite(loss)
        if algo == "svgd":
            results.append(opt.particles.detach().clone())
        elif algo == "ivon":
            results.append(flat([opt.state[p]["mean"] for p in model.parameters()]))
        else:
            results.append(flat(list(model.parameters())))
    if algo == "svgd":
        # the reference unscales only the LAST particle's gradients once per step through the base optimizer's
        # GradScaler state machine; what must hold is that the scaled run is finite and moves the particles
        assert torch.isfinite(results[1]).all()
    else:
        assert torch.allclose(results[0], results[1], rtol=1e-4, atol=1e-6), (results[0] - results[1]).abs().max()


def test_last_layer_combinator(backend):
    """LastLayerBayesianOptimizer (algo.py:83-105): Bayesian head + deterministic backbone."""
    ops, dev = backend
    torch.manual_seed(0)
    model = make_mlp().to(dev)
    head, body = list(model[2].parameters()), list(model[0].parameters())
    ll = bde.SwagOptimizer(head, torch.optim.SGD(head, lr=0.1), update_interval=1, deviation_samples=2, _ops=ops)
    det = torch.optim.SGD(body, lr=0.1)
    opt = bde.LastLayerBayesianOptimizer(ll, det)
    x, y = torch.randn(8, 13, device=dev), torch.randn(8, 1, device=dev)
    before = [p.detach().clone() for p in body + head]
    loss = opt.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward())
    assert torch.isfinite(loss)
    assert all(not torch.equal(a, b.detach()) for a, b in zip(before, body + head))     # both parts moved
    opt.complete_epoch()
    opt.sample_parameters()
    assert ll.state["__params_dirty"] and ll.state["__updates"] == 1
    sd = opt.state_dict()
    assert set(sd) == {"ll_bayesian_optimizer", "deterministic_optimizer"}
    with pytest.raises(RuntimeError):
        opt.get_base_optimizer()


def test_rbf_function(golden, backend):
    """rbf(particles) -> (kernel, grad_kernel), the drop-in for svgd.py:14-32."""
    ops, dev = backend
    g = golden("svgd_phi.npz")
    for i in (0, 3, 9):
        P = T(g[f"P_{i}"]).to(dev)
        K, gK = bde.rbf(P, _ops=ops)
        k64, gk64 = g[f"K64_{i}"], None
        assert np.max(np.abs(K.cpu().numpy() - k64)) <= 5e-6
        ref = g[f"gradK_{i}"]
        assert np.max(np.abs(gK.cpu().numpy() - ref)) <= 2e-5 * np.max(np.abs(ref)) + 1e-7


# ------------------------------------------------ reference checkpoints --
import os as _os

_GOLD = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")


def test_load_reference_swag_checkpoint(backend):
    """A state_dict written by the REFERENCE SwagOptimizer ([D] CPU mean, [D, K] rolled deviations, pickled
    base optimizer) loads into the flat / ring layout and resumes with the reference's counters."""
    ops, dev = backend
    ck = torch.load(_os.path.join(_GOLD, "ref_swag_checkpoint.pt"), weights_only=False)
    p1, p2 = nn.Parameter(ck["params"][0].clone().to(dev)), nn.Parameter(ck["params"][1].clone().to(dev))
    base = torch.optim.SGD([p1, p2], lr=0.1, momentum=0.9)
    opt = bde.SwagOptimizer([p1, p2], base, update_interval=2, start_epoch=0, deviation_samples=4, _ops=ops)
    sd = ck["optimizer"]
    sd["state"]["__base_optimizer"] = base        # the caller owns the base optimizer (reference: pickled along)
    opt.load_state_dict(sd)
    assert opt.state["__updates"] == 7 and opt.state["__epoch"] == 1 and opt.state["__steps_since_swag_start"] == 14
    np.testing.assert_array_equal(opt.mean_vector().cpu().numpy(), ck["mean"].numpy())
    np.testing.assert_array_equal(opt.sq_vector().cpu().numpy(), ck["sq"].numpy())
    np.testing.assert_array_equal(opt.deviations_dk().cpu().numpy(), ck["dev"].numpy())
    # ...and keeps going: two more steps = one more update, written to ring row 0 (= the oldest column)
    c1, c2 = ck["c"][0].to(dev), ck["c"][1].to(dev)
    for _ in range(2):
        opt.step(lambda: (p1 * c1).sum() + (p2 * c2).sum(), lambda l: l.backward())
    assert opt.state["__updates"] == 8 and opt.state["__dev_head"] == 1
    st = O_state_from_checkpoint(ck)
    theta = flat([p1, p2]).cpu()
    import oracle.bde_oracle as O
    st.updates = 8
    O.swag_moment_update(st, theta)
    np.testing.assert_array_equal(opt.mean_vector().cpu().numpy(), st.mean.numpy())
    np.testing.assert_array_equal(opt.deviations_dk().cpu().numpy(), st.deviations.numpy())
    # round trip through our own state_dict (reference wire layout)
    out = opt.state_dict()["state"]
    assert tuple(out["__deviations"].shape) == (13, 4) and out["__mean"].device.type == "cpu"


def O_state_from_checkpoint(ck):
    import oracle.bde_oracle as O
    return O.SwagState(mean=ck["mean"].clone(), sq_weights=ck["sq"].clone(), deviations=ck["dev"].clone(),
                       epoch=1, steps_since_swag_start=14, updates=7, column_iterate=[0, 0, 0, 0])


def test_load_reference_svgd_checkpoint_and_resume(backend):
    """Per-tensor particle_i entries written by the REFERENCE SVGDOptimizer are copied into the flat particle
    buffer; the next step reproduces the reference's next step."""
    ops, dev = backend
    ck = torch.load(_os.path.join(_GOLD, "ref_svgd_checkpoint.pt"), weights_only=False)
    nxt = torch.load(_os.path.join(_GOLD, "ref_svgd_checkpoint_next.pt"), weights_only=False)
    model = make_mlp().to(dev)
    model.load_state_dict(ck["model"])
    ref_base = ck["optimizer"]["state"]["__base_optimizer"]            # the reference's pickled torch.optim.SGD
    base = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    opt = bde.SVGDOptimizer(model.parameters(), lambda: None, base, particle_count=3, dataset_size=64, l2_reg=0.01,
                            _ops=ops)
    sd = ck["optimizer"]
    sd["state"]["__base_optimizer"] = base
    opt.load_state_dict(sd)
    np.testing.assert_array_equal(opt.particles.cpu().numpy(), ck["particles"].numpy())
    # carry the shared momentum buffers over from the reference's base optimizer (keyed on ITS parameters)
    for p_new, p_old in zip(model.parameters(), ref_base.param_groups[0]["params"]):
        base.state[p_new]["momentum_buffer"] = ref_base.state[p_old]["momentum_buffer"].clone().to(dev)
    x, y = ck["x"].to(dev), ck["y"].to(dev)
    loss = opt.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward())
    assert abs(float(loss) - nxt["loss"]) <= 2e-5 * abs(nxt["loss"])
    np.testing.assert_allclose(opt.particles.cpu().numpy(), nxt["particles_after"].numpy(), rtol=2e-5, atol=3e-6)


# ------------------------------------------------------ integration (CNN) --
class _SmallCNN(nn.Module):
    def __init__(self, conv=None, linear=None):
        super().__init__()
        self.c1 = conv(3, 8, 3) if conv else nn.Conv2d(3, 8, 3, padding=1)
        self.bn = nn.BatchNorm2d(8)
        self.c2 = nn.Conv2d(8, 8, 3, padding=1, bias=False)
        self.fc = linear(8, 5) if linear else nn.Linear(8, 5)

    def forward(self, x):
        x = F.relu(self.bn(self.c1(x)))
        x = F.relu(self.c2(x)).mean(dim=(2, 3))
        return F.log_softmax(self.fc(x), dim=1)


@pytest.mark.parametrize("algo", ["svgd", "svgd_fused", "swag", "ivon", "bbb",
                                  pytest.param("bbb_fused_conv", marks=unverified("conv_lrt")),
                                  pytest.param("svgd_fused_small", marks=unverified("svgd_small", "small_step_host", "mean_scalars", "fast_loop"))])
def test_cnn_training_loop_like_the_reference_drivers(backend, algo):
    """The call sequence of the reference's train/eval loops (cifar.py:160-176, ensemble.py:28-44) on a small
    conv net with BatchNorm: 4-D parameters, a bias-free conv, buffers that are not parameters."""
    ops, dev = backend
    torch.manual_seed(0)
    prior = bde.GaussianPrior(0, 1.0)
    conv_kw = {"fused_conv": True} if algo == "bbb_fused_conv" else {}
    small_kw = SMALL if algo == "svgd_fused_small" else {}
    if algo.startswith("bbb"):
        algo = "bbb"
        model = _SmallCNN(conv=lambda i, o, k: bde.BBBConv2d(i, o, k, prior, prior, padding=1, _ops=ops, **conv_kw),
                          linear=lambda i, o: bde.BBBLinear(i, o, prior, prior, _ops=ops)).to(dev)
    else:
        model = _SmallCNN().to(dev)
    params = list(model.parameters())
    if algo.startswith("svgd"):
        base = torch.optim.SGD(params, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4)
        opt = bde.SVGDOptimizer(params, lambda: bde.reset_model_params(model), base, particle_count=3, dataset_size=64,
                                l2_reg=1e-5, fuse_base_optimizer=algo.startswith("svgd_fused"), reuse_gram=(algo == "svgd_fused"),
                                _ops=ops, **small_kw)
    elif algo == "swag":
        base = torch.optim.SGD(params, lr=0.05, momentum=0.9)
        opt = bde.SwagOptimizer(params, base, update_interval=2, start_epoch=1, deviation_samples=3, rng="philox", _ops=ops)
    elif algo == "ivon":
        opt = bde.iVONOptimizer(params, lr=1e-2, prior_prec=50.0, dataset_size=64, mc_samples=2, damping=1e-3,
                                rng="philox", _ops=ops)
        base = opt
    else:
        base = torch.optim.Adam(params, lr=1e-3)
        opt = bde.BBBOptimizer(params, base, prior, dataset_size=64, mc_samples=2, kl_rescaling=0.2, _ops=ops)
    sched = torch.optim.lr_scheduler.StepLR(opt.get_base_optimizer(), step_size=1, gamma=0.5)
    ens = bde.DeepEnsemble([(model, opt)])
    x = torch.randn(64, 3, 8, 8, generator=torch.Generator().manual_seed(1)).to(dev)
    y = torch.randint(0, 5, (64,), generator=torch.Generator().manual_seed(2)).to(dev)
    before = flat(params).clone()
    losses = []
    for epoch in range(3):
        model.train()
        for b in range(4):
            xb, yb = x[b * 16:(b + 1) * 16], y[b * 16:(b + 1) * 16]
            loss = opt.step(lambda: F.nll_loss(model(xb), yb), lambda l: l.backward())
            losses.append(float(loss.detach()))
        opt.complete_epoch()
        sched.step()
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] * 1.5
    assert not torch.equal(before, flat(params))
    model.eval()
    with torch.no_grad():
        out = ens.predict(lambda m: m(x[:8]), 6)
    assert out.shape == (6, 8, 5) and torch.isfinite(out).all()
    if algo in ("swag", "ivon", "svgd", "svgd_fused"):
        assert not torch.equal(out[0], out[1])              # different posterior samples / particles
    # state_dict round trip through the ensemble container (ensemble.py:17-26)
    sd = ens.state_dict()
    assert set(sd) == {"models", "optimizers"} and len(sd["optimizers"]) == 1
    # training continues after evaluation (SWAG restores its weights, swag.py:38)
    model.train()
    loss = opt.step(lambda: F.nll_loss(model(x[:16]), y[:16]), lambda l: l.backward())
    assert torch.isfinite(loss)


def test_svgd_fused_reuse_gram_tracks_unfused_over_many_steps(backend):
    """30 steps: the one-pass path (fused optimizer + Gram carried from step to step) stays on the trajectory of
    the three-launch path with the stock torch optimizer (no drift from reusing the previous kernel's Gram)."""
    ops, dev = backend
    g = torch.Generator().manual_seed(4)
    x, y = torch.randn(32, 13, generator=g).to(dev), torch.randn(32, 1, generator=g).to(dev)
    runs = []
    for fused in (False, True):
        torch.manual_seed(12)
        model = make_mlp().to(dev)
        base = torch.optim.SGD(model.parameters(), lr=0.02, momentum=0.9, nesterov=True, weight_decay=1e-4)
        opt = bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model), base, particle_count=6,
                                dataset_size=32, l2_reg=1e-4, fuse_base_optimizer=fused, reuse_gram=fused, _ops=ops)
        sched = torch.optim.lr_scheduler.StepLR(base, step_size=10, gamma=0.5)     # LR schedule reaches the fused kernel
        import warnings as _w
        with _w.catch_warnings():
            _w.simplefilter("error")             # incl. torch's "lr_scheduler.step() before optimizer.step()" (fused: the
            for t in range(30):                  # base optimizer steps inside the kernel and says so)
                xb, yb = x[(t % 2) * 16:(t % 2 + 1) * 16], y[(t % 2) * 16:(t % 2 + 1) * 16]
                opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())
                sched.step()
        runs.append(opt.particles.detach().cpu().clone())
    assert torch.isfinite(runs[1]).all()
    np.testing.assert_allclose(runs[1].numpy(), runs[0].numpy(), rtol=2e-3, atol=2e-5)


class _ScaledLinear(nn.Module):
    """Test helper, not product: a caller of ``GaussianParameter.sample()`` shaped like the reference's Rank1Linear
    (rank1.py:9-80, out of scope per SURVEY section 2) -- ``y = W (x * s) * r + b`` with ``components`` (s, r, b) sets used
    round robin -- so that its parameter names match the fixture written from the reference's layer."""

    def __init__(self, n_in, n_out, components, ops):
        super().__init__()
        self.layer = nn.Linear(n_in, n_out, bias=False)
        self.s = nn.ModuleList([bde.GaussianParameter(n_in, _ops=ops) for _ in range(components)])
        self.r = nn.ModuleList([bde.GaussianParameter(n_out, _ops=ops) for _ in range(components)])
        self.bias = nn.Parameter(torch.zeros((components, n_out)))
        self.components, self.component_counter = components, 0

    def forward(self, x):
        c = self.component_counter
        out = self.layer(x * self.s[c].sample()) * self.r[c].sample() + self.bias[c]
        self.component_counter = (c + 1) % self.components
        return out


def test_bbb_components_and_sample_callers_reproduce_reference_trajectory(golden, backend):
    """BBBOptimizer(components=2, l2_scale) (bbb.py:75-80) over layers that call GaussianParameter.sample() per forward
    (util.py:170-171), against the trajectory of the reference's BBBOptimizer over its own rank-1 layers, draws replayed."""
    ops, dev = backend
    g = golden("rank1.npz")
    tape = [T(g[f"eps_{i}"]) for i in range(int(g["n_eps"]))]
    prior = bde.GaussianPrior(0, 1.0)
    model = nn.Sequential(_ScaledLinear(13, 20, 2, ops), nn.ReLU(), _ScaledLinear(20, 1, 2, ops)).to(dev)
    for m in model.modules():
        if isinstance(m, bde.GaussianParameter):
            m.noise_source = lambda rho: tape.pop(0).to(rho.device)
    names = [str(n) for n in g["names"]]
    named = dict(model.named_parameters())
    assert set(named) == set(names)
    with torch.no_grad():
        for n in names:
            named[n].copy_(T(g[f"init/{n}"]).to(dev))
    params = [named[n] for n in names]
    opt = bde.BBBOptimizer(params, torch.optim.Adam(params, lr=5e-3), prior, dataset_size=32, mc_samples=1,
                           kl_rescaling=1.0, components=2, l2_scale=1e-2, _ops=ops)
    x, y = T(g["x"]).to(dev), T(g["y"]).to(dev)
    for t in range(4):
        xb, yb = x[(t % 2) * 16:(t % 2 + 1) * 16], y[(t % 2) * 16:(t % 2 + 1) * 16]
        loss = opt.step(lambda: sum(F.mse_loss(model(xb), yb) for _ in range(2)), lambda l: l.backward())
        assert abs(float(loss.detach()) - g["losses"][t]) <= 1e-5 * abs(g["losses"][t])
        np.testing.assert_allclose(flat(params).cpu().numpy(), g["traj"][t], rtol=2e-4, atol=2e-5)
    assert not tape and model[0].component_counter == 0


# ------------------------------------------- BBB round-2 fixtures (bbb2.npz) --
def _load_named(g, tag, named, dev):
    names = [str(n) for n in g[f"{tag}_names"]]
    with torch.no_grad():
        for n in names:
            named[n].copy_(T(g[f"{tag}_init/{n}"]).to(dev))
    return [named[n] for n in names]


def test_bbb_mixture_prior_trajectory(golden, backend):
    """BBBOptimizer with the reference's MixturePrior (bbb.py:23-37): the means take the autograd path, and the
    optimizer-level zero_grad (bbb.py:60) must clear the rho gradients every step -- they only receive the data
    term, so a missed clear shows up as a drifting trajectory (4 steps, SGD with momentum)."""
    ops, dev = backend
    MixturePrior = bde.MixturePrior              # exported like the reference's src.algos.bbb.MixturePrior (bbb.py:23)
    g = golden("bbb2.npz")
    tape = [T(g[f"d_eps_{i}"]) for i in range(int(g["d_n_eps"]))]
    model = nn.Sequential(SampledLinear(13, 20, tape, ops), nn.ReLU(), SampledLinear(20, 1, tape, ops)).to(dev)
    extra = nn.Parameter(torch.zeros(4, device=dev))
    named = dict(model.named_parameters())
    named["extra"] = extra
    params = _load_named(g, "d", named, dev)
    prior = MixturePrior(0.5, 1.0, 0.05)
    opt = bde.BBBOptimizer(params, torch.optim.SGD(params, lr=0.05, momentum=0.9), prior, dataset_size=48, mc_samples=2,
                           kl_rescaling=0.5, l2_scale=0.3, _ops=ops)
    x, y = T(g["d_x"]).to(dev), T(g["d_y"]).to(dev)
    for t in range(4):
        xb, yb = x[(t % 3) * 16:(t % 3 + 1) * 16], y[(t % 3) * 16:(t % 3 + 1) * 16]
        loss = opt.step(lambda: F.mse_loss(model(xb), yb) + extra.sum() * 0.01, lambda l: l.backward())
        want = g["d_losses"][t]
        assert abs(float(loss) - want) <= 1e-5 * abs(want), (t, float(loss), want)
        np.testing.assert_allclose(flat(params).cpu().numpy(), g["d_traj"][t], rtol=1e-4, atol=2e-5)
    assert not tape


def test_bbb_frozen_parameters_do_not_move(golden, backend):
    """requires_grad=False parameters handed to BBBOptimizer (a Gaussian mean, a rho, a plain tensor): in the
    reference their .grad stays None, so the base optimizer -- weight decay included -- skips them; the trainable
    rest follows the reference's trajectory."""
    ops, dev = backend
    g = golden("bbb2.npz")
    tape = [T(g[f"e_eps_{i}"]) for i in range(int(g["e_n_eps"]))]
    model = nn.Sequential(SampledLinear(13, 20, tape, ops), nn.ReLU(), SampledLinear(20, 1, tape, ops)).to(dev)
    model[0].bias.mean.requires_grad_(False)
    model[2].weight.rho.requires_grad_(False)
    extra = nn.Parameter(torch.zeros(4, device=dev))
    frozen_plain = nn.Parameter(torch.zeros(6, device=dev), requires_grad=False)
    named = dict(model.named_parameters())
    named["extra"], named["frozen_plain"] = extra, frozen_plain
    params = _load_named(g, "e", named, dev)
    before = {n: named[n].detach().clone() for n in ("0.bias.mean", "2.weight.rho", "frozen_plain")}
    prior = bde.GaussianPrior(0, 1.0)
    base = torch.optim.SGD(params, lr=0.05, momentum=0.9, weight_decay=0.1)
    opt = bde.BBBOptimizer(params, base, prior, dataset_size=48, mc_samples=1, kl_rescaling=1.0, l2_scale=1.0, _ops=ops)
    x, y = T(g["e_x"]).to(dev), T(g["e_y"]).to(dev)
    for t in range(3):
        xb, yb = x[t * 16:(t + 1) * 16], y[t * 16:(t + 1) * 16]
        loss = opt.step(lambda: F.mse_loss(model(xb), yb) + extra.sum() * 0.01, lambda l: l.backward())
        want = g["e_losses"][t]
        assert abs(float(loss) - want) <= 1e-5 * abs(want), (t, float(loss), want)
        np.testing.assert_allclose(flat(params).cpu().numpy(), g["e_traj"][t], rtol=1e-4, atol=2e-5)
        for n, v in before.items():
            assert torch.equal(named[n].detach(), v), n
            assert named[n].grad is None, n


class _BdeCNN(nn.Module):
    def __init__(self, prior, ops, conv_kw):
        super().__init__()
        self.conv = bde.BBBConv2d(3, 4, 3, prior, prior, padding=1, _ops=ops, **conv_kw)
        self.conv2 = bde.BBBConv2d(4, 4, 3, prior, prior, stride=2, bias=False, _ops=ops, **conv_kw)
        self.fc = bde.BBBLinear(4, 2, prior, prior, _ops=ops)

    def forward(self, x):
        x = F.relu(self.conv(x))
        x = F.relu(self.conv2(x)).mean(dim=(2, 3))
        return self.fc(x)


@pytest.mark.parametrize("element_wise", ["native_below_threshold", "fused_passes"])
def test_bbb_conv_layers_reproduce_reference_cnn_trajectory(golden, backend, monkeypatch, element_wise, conv_kw):
    """bde.BBBConv2d (padding / stride / bias-free) + bde.BBBLinear under BBBOptimizer against the trajectory of the
    REFERENCE's BBBConv2d + BBBLinear + BBBOptimizer on the same small CNN (bbb_layers.py:105-159), noise replayed.
    Once as a layer of this size runs (element-wise pieces as native ATen nodes) and once with the size threshold at
    zero, so that the fused variance-operand and epilogue passes (bde_var_operand_*, bde_local_reparam_*) carry it."""
    ops, dev = backend
    import beyond_deep_ensembles_amd.bbb_layers as L
    if element_wise == "fused_passes":
        monkeypatch.setattr(L, "_FUSE_MIN_ELEMS", 0)
    g = golden("bbb2.npz")
    tape = [T(g[f"f_eps_{i}"]) for i in range(int(g["f_n_eps"]))]
    monkeypatch.setattr(L, "normal_like", lambda t: tape.pop(0).to(t.device))
    prior = bde.GaussianPrior(0, 1.0)
    model = _BdeCNN(prior, ops, conv_kw).to(dev)
    params = _load_named(g, "f", dict(model.named_parameters()), dev)
    opt = bde.BBBOptimizer(params, torch.optim.Adam(params, lr=1e-2), prior, dataset_size=32, mc_samples=2,
                           kl_rescaling=0.2, _ops=ops)
    x, y = T(g["f_x"]).to(dev), T(g["f_y"]).to(dev)
    for t in range(3):
        xb, yb = x[(t % 2) * 16:(t % 2 + 1) * 16], y[(t % 2) * 16:(t % 2 + 1) * 16]
        loss = opt.step(lambda: F.mse_loss(model(xb), yb), lambda l: l.backward())
        want = g["f_losses"][t]
        assert abs(float(loss.detach()) - want) <= 1e-5 * abs(want), (t, float(loss), want)
        np.testing.assert_allclose(flat(params).cpu().numpy(), g["f_traj"][t], rtol=2e-4, atol=3e-5)
    assert not tape


@unverified("conv_lrt")
def test_bbb_conv2d_fused_path_selection_and_weight_cache(backend, monkeypatch):
    """BBBConv2d takes the fused op (bde_conv_lrt_*) in training mode for fp32 NCHW inputs and supported geometries, the
    stock convolutions otherwise (eval-mode frozen noise, padding='same', fused_conv=False); the prepared weight buffer is
    computed once per weight VERSION (mc_samples forward / backward passes share it), again after an optimizer step, an
    in-place edit, or -- for writes through .data, which bump no version counter -- BBBOptimizer.step's epoch /
    layer.invalidate_sigma_cache(); fused and stock paths agree (same noise), forward and all five gradients."""
    import beyond_deep_ensembles_amd.bbb_layers as BL
    ops, dev = backend
    torch.manual_seed(5)
    prior = bde.GaussianPrior(0, 1.0)
    conv = bde.BBBConv2d(5, 7, 3, prior, prior, stride=2, padding=1, fused_conv=True, _ops=ops).to(dev)
    ref = bde.BBBConv2d(5, 7, 3, prior, prior, stride=2, padding=1, fused_conv=False, _ops=ops).to(dev)
    ref.load_state_dict(conv.state_dict())
    x = torch.randn(3, 5, 9, 11, device=dev, requires_grad=True)
    preps, fwds = [], []
    real_prep, real_fwd = ops.conv_lrt_prep, ops.conv_lrt_fwd
    ops.conv_lrt_prep = lambda *a, **k: (preps.append(1), real_prep(*a, **k))[1]
    ops.conv_lrt_fwd = lambda *a, **k: (fwds.append(1), real_fwd(*a, **k))[1]
    native = BL._native_nodes(ops)                    # on the device the forward goes through the C++ node instead
    if native is not None and hasattr(native, "conv_lrt"):
        real_node = native.conv_lrt
        monkeypatch.setattr(native, "conv_lrt", lambda *a, **k: (fwds.append(1), real_node(*a, **k))[1])
    try:
        noise = torch.randn(3, 7, 5, 6)
        real_normal_like = BL.normal_like
        monkeypatch.setattr(BL, "normal_like", lambda t: noise.to(t.device))
        outs = {}
        for name, layer in (("fused", conv), ("stock", ref)):
            leaves = [x, layer.weight.mean, layer.weight.rho, layer.bias.mean, layer.bias.rho]
            out = layer(x)
            outs[name] = [out.detach()] + list(torch.autograd.grad(out.pow(2).sum(), leaves))
        assert len(fwds) == 1 and len(preps) == 1
        for a, b in zip(outs["fused"], outs["stock"]):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-5)
        monkeypatch.setattr(BL, "normal_like", real_normal_like)
        conv(x), conv(x)                                             # same weight version: no new preparation
        assert len(preps) == 1 and len(fwds) == 3
        with torch.no_grad():
            conv.weight.rho.add_(0.1)                                # an in-place change is a new version
        conv(x)
        assert len(preps) == 2
        conv.weight.rho.data.fill_(-2.5)                             # .data writes keep address and version ...
        conv(x)
        assert len(preps) == 2
        conv.invalidate_sigma_cache()                                # ... the layer's own invalidation catches them
        conv(x)
        assert len(preps) == 3
        n = len(fwds)
        conv.eval()
        conv(x)                                                      # frozen noise: the stock path
        assert len(fwds) == n
        conv.train()
        same = bde.BBBConv2d(5, 7, 3, prior, prior, padding="same", fused_conv=True, _ops=ops).to(dev)
        assert same(x).shape == (3, 7, 9, 11) and len(fwds) == n     # padding='same': stock convolutions
    finally:
        ops.conv_lrt_prep, ops.conv_lrt_fwd = real_prep, real_fwd


def test_r5_bayesian_layers_pickle_and_deepcopy(backend, monkeypatch, conv_kw):
    """The reference's layers are plain nn.Modules: picklable, deep-copyable (a closure stored on the layer broke
    both).  A copy starts with cold caches of its own, and invalidate_sigma_cache() of the copy drops the COPY's caches."""
    import copy
    import io
    import pickle
    import beyond_deep_ensembles_amd.algo as A
    ops, dev = backend
    monkeypatch.setattr(A, "_default_ops", lambda: ops)     # a copy binds the backend again at its first use: this one
    torch.manual_seed(2)
    prior = bde.GaussianPrior(0, 1.0)
    lin = bde.BBBLinear(1024, 1100, prior, prior, _ops=ops).to(dev)          # wide enough for the sigma^2 cache
    conv = bde.BBBConv2d(3, 4, 3, prior, prior, padding=1, _ops=ops, **conv_kw).to(dev)
    x_lin, x_conv = torch.randn(4, 1024, device=dev), torch.randn(2, 3, 6, 6, device=dev)
    lin(x_lin), conv(x_conv)                                                 # fills the caches where the backend has them
    for layer, x in ((lin, x_lin), (conv, x_conv)):
        for clone in (copy.deepcopy(layer), pickle.loads(pickle.dumps(layer))):
            assert clone._sigma_cache is not layer._sigma_cache and clone._conv_weights is not layer._conv_weights
            assert clone._sigma_cache.key is None and clone._conv_weights.key is None
            for a, b in zip(clone.state_dict().values(), layer.state_dict().values()):
                assert torch.equal(a, b)
            assert clone(x).shape == layer(x).shape
            keys = (layer._sigma_cache.key, layer._conv_weights.key)
            clone.invalidate_sigma_cache()
            assert clone._sigma_cache.key is None and clone._conv_weights.key is None
            assert (layer._sigma_cache.key, layer._conv_weights.key) == keys          # the original's caches are untouched
        buf = io.BytesIO()
        torch.save(layer, buf)                                               # whole-module save, as torch.save(model) does
        buf.seek(0)
        assert isinstance(torch.load(buf, weights_only=False), type(layer))


@unverified("conv_lrt")
def test_r5_fused_conv_auto_follows_the_measured_table(backend, monkeypatch, tmp_path):
    """fused_conv="auto" (the default): the fused kernels only where conv_profit.json holds a device measurement of this
    kernel version that beats the stock sequence for the pass at hand; no record, another ABI version, a speed-up below 1
    or a much smaller batch -> the stock convolutions.  True forces, False forbids."""
    import json
    import beyond_deep_ensembles_amd.bbb_layers as BL
    from beyond_deep_ensembles_amd import conv_profit
    ops, dev = backend
    prior = bde.GaussianPrior(0, 1.0)
    calls = []
    real_fwd = ops.conv_lrt_fwd
    ops.conv_lrt_fwd = lambda *a, **k: (calls.append(1), real_fwd(*a, **k))[1]
    native = BL._native_nodes(ops)
    if native is not None and hasattr(native, "conv_lrt"):
        real_node = native.conv_lrt
        monkeypatch.setattr(native, "conv_lrt", lambda *a, **k: (calls.append(1), real_node(*a, **k))[1])
    abi = int(ops.lib.bde_version()) if hasattr(getattr(ops, "lib", None), "bde_version") else -1
    try:
        layer = bde.BBBConv2d(3, 4, 3, prior, prior, padding=1, _ops=ops).to(dev)
        assert layer.fused_conv == "auto"
        x = torch.randn(8, 3, 6, 6, device=dev)

        def used(table, inp=x, grad=True):
            monkeypatch.setattr(conv_profit, "_table", table)
            n = len(calls)
            with torch.enable_grad() if grad else torch.no_grad():
                layer(inp)
            return len(calls) - n
        empty = {"abi": abi, "source": "", "layers": {}}
        assert used(empty) == 0
        key = conv_profit._key(3, 4, 3, 1, 1, 6, 6)
        win = {"abi": abi, "source": "t", "layers": {key: {"batch": 8, "fwd": 2.0, "fwd_bwd": 1.3}}}
        assert used(win) == 1 and used(win, grad=False) == 1
        fwd_only = {"abi": abi, "source": "t", "layers": {key: {"batch": 8, "fwd": 2.0, "fwd_bwd": 0.8}}}
        assert used(fwd_only) == 0 and used(fwd_only, grad=False) == 1      # training passes keep the stock path
        assert used(dict(win, abi=abi + 1)) == 0                              # measured with other kernels: unmeasured
        big_batch = {"abi": abi, "source": "t", "layers": {key: {"batch": 128, "fwd": 2.0, "fwd_bwd": 2.0}}}
        assert used(big_batch) == 0                                           # measured at 128 images, asked for 8
        assert used(win, inp=torch.randn(8, 3, 7, 6, device=dev)) == 0       # another image size: no record
        layer.fused_conv = True
        assert used(empty) == 1
        layer.fused_conv = False
        assert used(win) == 0
        path = tmp_path / "conv_profit.json"
        path.write_text(json.dumps(win))
        assert conv_profit.load(str(path))["layers"][key]["fwd"] == 2.0
        path.write_text("not json")
        assert conv_profit.load(str(path))["layers"] == {}
        # a record with autotuned tilings: pinned through the library's hooks the first time the layer runs (backends with kernels)
        if hasattr(ops, "conv_lrt_candidates"):
            xs, ws_ = (8, 3, 6, 6), (4, 3, 3, 3)
            geo = ops.conv_lrt_pass_geos(0, xs, ws_, (1, 1), (1, 1))[0]
            cands, chosen = ops.conv_lrt_candidates(geo)
            other = next(i for i in range(len(cands)) if i != chosen)
            wc, wchosen = ops.conv_lrt_wgrad_candidates(xs, ws_, (1, 1), (1, 1))
            tuned = {"abi": abi, "source": "t", "layers": {key: {"batch": 8, "fwd": 2.0, "fwd_bwd": 2.0, "tilings": {
                "launch": [list(geo) + list(cands[other][:4])], "wgrad": list(wc[-1][:4])}}}}
            layer.fused_conv = "auto"
            conv_profit._applied.clear()
            try:
                assert used(tuned) == 1
                assert ops.conv_lrt_candidates(geo)[1] == other
                assert ops.conv_lrt_wgrad_candidates(xs, ws_, (1, 1), (1, 1))[1] == len(wc) - 1
            finally:
                ops.conv_lrt_set_tiling(geo, None)
                ops.conv_lrt_wgrad_set_tiling(xs, ws_, (1, 1), (1, 1), None)
                conv_profit._applied.clear()
        shipped = conv_profit.load(conv_profit._PATH)                        # the shipped table parses and is well formed
        for rec in shipped["layers"].values():
            assert {"batch", "fwd", "fwd_bwd"} <= set(rec)
    finally:
        ops.conv_lrt_fwd = real_fwd


@unverified("conv_lrt")
def test_r5_fused_conv_forward_without_gradients_writes_no_variance(backend, monkeypatch):
    """The fused forward stores the total variance only for its backward (sqrt(var)): under torch.no_grad() -- or when nothing
    requires a gradient -- bde_conv_lrt_fwd gets var_out = NULL (one output-sized store less) and returns the same sample."""
    import beyond_deep_ensembles_amd.bbb_layers as BL
    ops, dev = backend
    if not hasattr(ops, "conv_lrt_fwd"):
        pytest.skip("backend without the fused convolution")
    torch.manual_seed(8)
    prior = bde.GaussianPrior(0, 1.0)
    layer = bde.BBBConv2d(3, 4, 3, prior, prior, padding=1, fused_conv=True, _ops=ops).to(dev)
    x = torch.randn(2, 3, 6, 6, device=dev)
    noise = torch.randn(2, 4, 6, 6)
    monkeypatch.setattr(BL, "normal_like", lambda t: noise.to(t.device))
    seen = []
    real = ops.conv_lrt_fwd

    def spy(x_, wbuf, w_shape, b_mu, bias_var, stride, padding, out, var_out, **kw):
        seen.append(var_out is not None)
        return real(x_, wbuf, w_shape, b_mu, bias_var, stride, padding, out, var_out, **kw)
    native = BL._native_nodes(ops)
    if native is not None and hasattr(native, "conv_lrt"):
        monkeypatch.setattr(BL, "_native_nodes", lambda o: None)       # the Python Function: its ops call can be observed
    ops.conv_lrt_fwd = spy
    try:
        with_grad = layer(x)
        with torch.no_grad():
            without = layer(x)
        assert seen == [True, False]
        assert torch.equal(with_grad.detach(), without)
    finally:
        ops.conv_lrt_fwd = real
    if native is not None and hasattr(native, "conv_lrt"):             # the C++ node takes the flag as an argument
        monkeypatch.setattr(BL, "_native_nodes", lambda o: native)
        with torch.no_grad():
            assert torch.equal(layer(x), without)
        assert torch.equal(layer(x).detach(), without)


def test_bbb_group_draw_is_one_launch_per_forward(backend):
    """With rng="philox" the Gaussian parameters owned by a BBBOptimizer are drawn by ONE bde_gauss_draw_fwd launch
    over the group's flat buffers per forward pass (and ONE bde_gauss_draw_bwd per backward), and the result is
    w = mean + softplus(rho) * eps with the Philox noise of the flat element index; gradients match autograd."""
    ops, dev = backend
    from oracle import philox as PH
    torch.manual_seed(4)

    class Lin(nn.Module):
        def __init__(self, i, o):
            super().__init__()
            self.weight = bde.GaussianParameter((o, i), rng="philox", seed=77, _ops=ops)
            self.bias = bde.GaussianParameter((o,), rng="philox", seed=77, _ops=ops)
            self.weight.blundell_init()
            self.bias.blundell_init()

        def forward(self, x):
            return F.linear(x, self.weight.sample(), self.bias.sample())

    model = nn.Sequential(Lin(13, 9), nn.Tanh(), Lin(9, 2)).to(dev)
    params = list(model.parameters())
    opt = bde.BBBOptimizer(params, torch.optim.SGD(params, lr=0.0), bde.GaussianPrior(0, 1.0), dataset_size=10,
                           mc_samples=2, _ops=ops)
    calls = {"fwd": 0, "bwd": 0, "streams": []}
    real_fwd, real_bwd = ops.gauss_draw_fwd, ops.gauss_draw_bwd

    def fwd(*a, **k):
        calls["fwd"] += 1
        calls["streams"].append(k.get("stream_id"))
        return real_fwd(*a, **k)

    def bwd(*a, **k):
        calls["bwd"] += 1
        return real_bwd(*a, **k)
    ops.gauss_draw_fwd, ops.gauss_draw_bwd = fwd, bwd
    try:
        x = torch.randn(5, 13, device=dev)
        drawn = []

        def forward():
            out = model(x)
            drawn.append([model[0].weight._flat_group._draw[i].detach().clone() for i in range(4)])
            return out.pow(2).mean()
        opt.step(forward, lambda l: l.backward())
    finally:
        ops.gauss_draw_fwd, ops.gauss_draw_bwd = real_fwd, real_bwd
    assert calls["fwd"] == 2 and calls["bwd"] == 2                     # mc_samples = 2 forward passes, 4 tensors each
    group = model[0].weight._flat_group
    means = [p.detach().cpu() for p in group.means]
    rhos = [p.detach().cpu() for p in group.rhos]
    d = sum(m.numel() for m in means)
    for draw, stream in zip(drawn, calls["streams"]):
        eps = torch.from_numpy(PH.normals(77, stream, d)).float()
        off = 0
        for w, m, r in zip(draw, means, rhos):
            e = eps[off:off + m.numel()].view(m.shape)
            off += m.numel()
            want = m + F.softplus(r) * e
            assert torch.allclose(w.cpu(), want, rtol=1e-5, atol=1e-6)
    # gradients: one more forward/backward by hand against torch autograd with the same noise
    for p in params:
        p.grad = None
    out = model(x).pow(2).mean()
    out.backward()
    stream = calls["streams"][-1] + 1
    eps = torch.from_numpy(PH.normals(77, stream, d)).float()
    ms = [m.clone().requires_grad_(True) for m in means]
    rs = [r.clone().requires_grad_(True) for r in rhos]
    off, ws = 0, []
    for m, r in zip(ms, rs):
        ws.append(m + F.softplus(r) * eps[off:off + m.numel()].view(m.shape))
        off += m.numel()
    xc = x.cpu()
    ref = F.linear(torch.tanh(F.linear(xc, ws[0], ws[1])), ws[2], ws[3]).pow(2).mean()
    ref.backward()
    assert abs(float(out) - float(ref)) <= 1e-5 * abs(float(ref))
    for p, m in zip(group.means, ms):
        assert torch.allclose(p.grad.cpu(), m.grad, rtol=1e-4, atol=1e-6)
    for p, r in zip(group.rhos, rs):
        assert torch.allclose(p.grad.cpu(), r.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_svgd_fused_state_lives_in_the_base_optimizer(backend, kind):
    """fuse_base_optimizer keeps the SHARED momentum / Adam moments in flat buffers -- and publishes them as
    base_optimizer.state[param] views, so base.state_dict() is complete and a run continued WITHOUT the fused path
    (or in the reference) carries on from the same state: fused 3 steps == fused 2 steps + 1 unfused step."""
    ops, dev = backend
    x = torch.randn(16, 13, generator=torch.Generator().manual_seed(8)).to(dev)
    y = torch.randn(16, 1, generator=torch.Generator().manual_seed(9)).to(dev)

    def make(fuse, seed=3):
        torch.manual_seed(seed)
        model = make_mlp().to(dev)
        base = (torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, nesterov=True) if kind == "sgd"
                else torch.optim.Adam(model.parameters(), lr=1e-2))
        opt = bde.SVGDOptimizer(model.parameters(), lambda: bde.reset_model_params(model), base, particle_count=3,
                                dataset_size=64, l2_reg=0.01, fuse_base_optimizer=fuse, single_launch=False, _ops=ops)
        return model, base, opt

    def step(model, opt):
        return opt.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward())

    model_a, base_a, opt_a = make(True)
    for _ in range(3):
        step(model_a, opt_a)
    model_b, base_b, opt_b = make(True)
    for _ in range(2):
        step(model_b, opt_b)
    params_b = list(model_b.parameters())
    sd = base_b.state_dict()                                   # complete: one entry per parameter
    assert len(sd["state"]) == len(params_b)
    key = "momentum_buffer" if kind == "sgd" else "exp_avg"
    assert all(key in e for e in sd["state"].values())
    if kind == "adam":
        assert all(float(e["step"]) == 6.0 for e in sd["state"].values())     # 2 steps x 3 particles (Q5)
    # continue unfused on the SAME base optimizer and particles
    opt_c = bde.SVGDOptimizer(params_b, lambda: None, base_b, particle_count=3, dataset_size=64, l2_reg=0.01,
                              single_launch=False, _ops=ops)
    with torch.no_grad():
        opt_c._P.copy_(opt_b._P)
    if kind == "adam":
        base_b.load_state_dict(sd)                             # per-parameter step counters as torch expects them
    step(model_b, opt_c)
    np.testing.assert_allclose(opt_c.particles.cpu().numpy(), opt_a.particles.cpu().numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.gpu
def test_native_autograd_nodes_equal_python_nodes():
    """The C++ autograd nodes of lib/_bde_host.so (csrc/host_autograd.cpp) and the Python Functions of bbb_layers.py are
    the same nodes: same kernels through the same C ABI, bit-identical outputs and gradients."""
    import beyond_deep_ensembles_amd.bbb_layers as L
    from beyond_deep_ensembles_amd.ops import HipOps
    ops, dev = HipOps(), torch.device("cuda:0")
    native = L._native_nodes(ops)
    assert native is not None, "lib/_bde_host.so is missing or does not load: run __graft_entry__.build()"
    check_native_nodes_equal_python_nodes(ops, dev, native)


def check_native_nodes_equal_python_nodes(ops, dev, native):
    """(also run by tests/test_hip_emu.py with host_autograd.cpp and the kernel sources built for the CPU model)"""
    import beyond_deep_ensembles_amd.bbb_layers as L
    torch.manual_seed(31)
    for b, i, o, bias in [(16, 2048, 182, True), (5, 13, 50, True), (70, 129, 33, False), (64, 1024, 1100, True)]:
        x = torch.randn(3, b // 3 + 1, i, device=dev)[:, : max(1, b // 3)]          # a batch with leading dimensions
        w_mu, w_rho = torch.randn(o, i, device=dev) * 0.1, torch.randn(o, i, device=dev) - 3
        b_mu, b_rho = (torch.randn(o, device=dev) * 0.1, torch.randn(o, device=dev) - 3) if bias else (None, None)
        for eps in (None, torch.randn(x.shape[:-1] + (o,), device=dev)):
            res = []
            for node in ("native", "python"):
                leaves = [t.clone().requires_grad_(True) if t is not None else None for t in (x, w_mu, w_rho, b_mu, b_rho)]
                if node == "native":
                    out = native.lrt_linear(*leaves, True, eps, 5, 9)
                else:
                    out = L._LrtLinear.apply(*leaves, True, eps, 5, 9, ops)
                grads = torch.autograd.grad(out, [t for t in leaves if t is not None], grad_outputs=torch.ones_like(out) * 0.5)
                res.append([out.detach()] + list(grads))
            assert all(torch.equal(a, c) for a, c in zip(*res)), (b, i, o, bias, eps is None)
    mean, var = torch.randn(4, 6, 5, 5, device=dev), torch.rand(4, 6, 5, 5, device=dev) + 0.1
    for eps in (None, torch.randn_like(mean)):
        res = []
        for node in ("native", "python"):
            m, v = mean.clone().requires_grad_(True), var.clone().requires_grad_(True)
            out = native.local_reparam(m, v, eps, 3, 4) if node == "native" else L._LocalReparam.apply(m, v, eps, 3, 4, ops)
            res.append([out.detach()] + list(torch.autograd.grad(out, [m, v], grad_outputs=torch.full_like(out, 2.0))))
        assert all(torch.equal(a, c) for a, c in zip(*res))
    for mode in (0, 1, 2):
        v0 = torch.randn(7, 11, device=dev) * 2
        res = []
        for node in ("native", "python"):
            v = v0.clone().requires_grad_(True)
            out = native.var_operand(v, mode) if node == "native" else L._VarOperand.apply(v, mode, ops)
            res.append([out.detach(), torch.autograd.grad(out, v, grad_outputs=torch.full_like(out, 0.3))[0]])
        assert all(torch.equal(a, c) for a, c in zip(*res)), mode
    # the fused convolution layer (round 4): C++ node == Python Function, with and without bias / supplied noise / input gradient
    for n, c, h, w, o, k, s_, p_, bias in [(4, 16, 32, 32, 16, 3, 1, 1, True), (3, 5, 9, 11, 7, 3, 2, 1, False), (2, 32, 16, 16, 64, 3, 2, 1, True)]:
        x0 = torch.randn(n, c, h, w, device=dev)
        w_mu, w_rho = torch.randn(o, c, k, k, device=dev) * 0.1, torch.randn(o, c, k, k, device=dev) - 3
        b_mu, b_rho = (torch.randn(o, device=dev) * 0.1, torch.randn(o, device=dev) - 3) if bias else (None, None)
        wbuf = ops.conv_lrt_wbuf(w_mu.shape, dev)
        ops.conv_lrt_prep(w_mu, w_rho, wbuf, b_rho, stride=(s_, s_), padding=(p_, p_))      # as the layer prepares it
        ho, wo = (h + 2 * p_ - k) // s_ + 1, (w + 2 * p_ - k) // s_ + 1
        for eps in (None, torch.randn(n, o, ho, wo, device=dev)):
            for x_grad in (True, False):
                res = []
                for node in ("native", "python"):
                    leaves = [t.clone().requires_grad_(True) if t is not None else None for t in (x0, w_mu, w_rho, b_mu, b_rho)]
                    leaves[0].requires_grad_(x_grad)
                    if node == "native":
                        out = native.conv_lrt(*leaves, s_, s_, p_, p_, eps, 5, 9, wbuf, True)
                    else:
                        out = L._ConvLrt.apply(*leaves, (s_, s_), (p_, p_), eps, 5, 9, ops, wbuf, True)
                    wrt = [t for t in leaves if t is not None and t.requires_grad]
                    grads = torch.autograd.grad(out, wrt, grad_outputs=torch.ones_like(out) * 0.5)
                    res.append([out.detach()] + list(grads))
                assert all(torch.equal(a, c2) for a, c2 in zip(*res)), (n, c, h, w, o, bias, eps is None, x_grad)
    # a layer without input gradient, and errors raised as exceptions
    xin = torch.randn(8, 13, device=dev)
    leaves = [t.requires_grad_(True) for t in (torch.randn(50, 13, device=dev), torch.randn(50, 13, device=dev) - 3)]
    out = native.lrt_linear(xin, leaves[0], leaves[1], None, None, True, None, 1, 2)
    assert all(g is not None for g in torch.autograd.grad(out.sum(), leaves))
    with pytest.raises(RuntimeError):
        native.lrt_linear(xin.double(), leaves[0], leaves[1], None, None, True, None, 1, 2)
    with pytest.raises(RuntimeError):
        native.lrt_linear(torch.randn(200, 13, device=dev), leaves[0], leaves[1], None, None, True, None, 1, 2)   # B > 128


@pytest.mark.parametrize("rows", ["up_to_128", "row_tiles", "above_128_default"])
def test_bbb_linear_layer_matches_reference_layer(golden, backend, monkeypatch, rows):
    """bde.BBBLinear forward + backward (fused ops: bde_lrt_linear_fwd / bde_lrt_linear_bwd on the GPU, their CPU
    restatement in the checker backend) against the REFERENCE's BBBLinear on the same seeded inputs (lrt.npz: output and
    all five gradients from the reference's autograd graph, at the UCI size, the iWildCam head, a wide layer).
    "row_tiles": lrt_tiled.npz, batches of 129 / 256 / 1000 rows with BBBLinear(fused_linear_max_rows=1024): ceil(rows / 128)
    launches of the same kernels, weight and bias gradients accumulated over the tiles (bbb_layers.py:61-80 at any batch);
    "above_128_default": the same batches through a default-constructed layer (two stock GEMMs + fused element-wise passes)."""
    ops, dev = backend
    import beyond_deep_ensembles_amd.bbb_layers as L
    from oracle.lrt_cases import lrt_case_inputs
    g = golden("lrt.npz" if rows == "up_to_128" else "lrt_tiled.npz")
    prior = bde.GaussianPrior(0, 1.0)
    fused_rows = []
    real_fwd = ops.lrt_linear_fwd
    monkeypatch.setattr(ops, "lrt_linear_fwd", lambda x, *a, **k: (fused_rows.append(x.shape[0]), real_fwd(x, *a, **k))[1])
    native = L._native_nodes(ops)
    if native is not None:
        real_node = native.lrt_linear
        monkeypatch.setattr(native, "lrt_linear", lambda x, *a, **k: (fused_rows.append(x.shape[0]), real_node(x, *a, **k))[1])
    for seed, b, i, o in g["cases"].tolist():
        x, w_mu, w_rho, b_mu, b_rho, eps, gout, probe = [T(a).to(dev) for a in lrt_case_inputs(seed, b, i, o)]
        kw = dict(fused_linear_max_rows=1024) if rows == "row_tiles" else {}
        layer = bde.BBBLinear(i, o, prior, prior, _ops=ops, **kw).to(dev).train()
        del fused_rows[:]
        with torch.no_grad():
            layer.weight.mean.copy_(w_mu)
            layer.weight.rho.copy_(w_rho)
            layer.bias.mean.copy_(b_mu)
            layer.bias.rho.copy_(b_rho)
        monkeypatch.setattr(L, "normal_like", lambda t: eps.to(t.device))
        xin = x.clone().requires_grad_(True)
        out = layer(xin)
        # which path ran: one fused launch, ceil(b / 128) of them, or none (the stock GEMMs)
        want_rows = {"up_to_128": [b], "above_128_default": []}.get(rows, [128] * (b // 128) + ([b % 128] if b % 128 else []))
        assert fused_rows == want_rows, (rows, b, fused_rows)
        leaves = [xin, layer.weight.mean, layer.weight.rho, layer.bias.mean, layer.bias.rho]
        gx, gwm, gwr, gbm, gbr = [t.detach().cpu().double() for t in torch.autograd.grad(out, leaves, grad_outputs=gout)]
        t = f"c{seed}_"

        def close(ours, want, scale, what):
            want = torch.as_tensor(np.asarray(want), dtype=torch.float64)
            tol = 2e-5 * max(float(scale), 1e-6)                      # fp32 GEMMs in a different summation order
            assert (torch.as_tensor(ours, dtype=torch.float64) - want).abs().max().item() <= tol, (what, b, i, o)
        close(out.detach().cpu(), g[t + "out"], np.abs(g[t + "out"]).max(), "out")
        close(gx, g[t + "g_x"], np.abs(g[t + "g_x"]).max(), "g_x")
        close(gbm, g[t + "g_bmu"], np.abs(g[t + "g_bmu"]).max(), "g_bmu")
        close(gbr, g[t + "g_brho"], max(np.abs(g[t + "g_brho"]).max(), 1e-3), "g_brho")
        pr = probe.cpu().double()
        for name, gw in (("g_wmu", gwm), ("g_wrho", gwr)):
            amax = float(g[t + name + "_absmax"])
            close(gw.sum(1), g[t + name + "_rowsum"], amax * np.sqrt(i), name + " row sums")
            close(gw.sum(0), g[t + name + "_colsum"], amax * np.sqrt(o), name + " column sums")
            close((gw * pr).sum(), g[t + name + "_proj"], amax * np.sqrt(i * o), name + " projection")
            assert abs(gw.abs().max().item() - amax) <= 2e-5 * amax, name


@pytest.mark.parametrize("path", [pytest.param("fused", marks=unverified("conv_lrt")), "stock"])
def test_bbb_conv2d_layer_matches_reference_layer(golden, backend, monkeypatch, path):
    """bde.BBBConv2d forward + backward against the REFERENCE's BBBConv2d on the same seeded inputs (conv_lrt.npz, written
    by oracle/gen_golden.py from the reference's own layer and autograd graph): output and all five gradients at the CIFAR
    ResNet-20 layer shapes (first layer, the three stages, both stride-2 transitions, a 1x1 shortcut) and two ragged
    geometries.  "fused": bde_conv_lrt_fwd / _bwd_data / _bwd_weight (HIP; the CPU checker runs torch convolutions behind
    the same autograd Function); "stock": fused_conv=False, round 3's composition."""
    ops, dev = backend
    import beyond_deep_ensembles_amd.bbb_layers as L
    from oracle.conv_cases import conv_case_inputs, conv_probe_w
    g = golden("conv_lrt.npz")
    prior = bde.GaussianPrior(0, 1.0)
    fused_calls = []
    if path == "fused":                                  # the fused op is reached through the Python Function or the C++ node
        real_fwd = ops.conv_lrt_fwd
        monkeypatch.setattr(ops, "conv_lrt_fwd", lambda *a, **k: (fused_calls.append(1), real_fwd(*a, **k))[1])
        native = L._native_nodes(ops)
        if native is not None and hasattr(native, "conv_lrt"):
            real_node = native.conv_lrt
            monkeypatch.setattr(native, "conv_lrt", lambda *a, **k: (fused_calls.append(1), real_node(*a, **k))[1])
    for seed, n, c, h, w, o, k, stride, padding, bias in g["cases"].tolist():
        x, w_mu, w_rho, b_mu, b_rho, eps, gout, probe_x = [T(a).to(dev) for a in conv_case_inputs(seed, n, c, h, w, o, k, stride, padding)]
        layer = bde.BBBConv2d(c, o, k, prior, prior, stride=stride, padding=padding, bias=bool(bias), fused_conv=path == "fused",
                              _ops=ops).to(dev).train()
        with torch.no_grad():
            layer.weight.mean.copy_(w_mu)
            layer.weight.rho.copy_(w_rho)
            if bias:
                layer.bias.mean.copy_(b_mu)
                layer.bias.rho.copy_(b_rho)
        monkeypatch.setattr(L, "normal_like", lambda t: eps.to(t.device))
        xin = x.clone().requires_grad_(True)
        out = layer(xin)
        leaves = [xin, layer.weight.mean, layer.weight.rho] + ([layer.bias.mean, layer.bias.rho] if bias else [])
        grads = [t.detach().cpu().double() for t in torch.autograd.grad(out, leaves, grad_outputs=gout)]
        t = f"c{seed}_"
        case = (seed, n, c, h, w, o, k, stride, padding, bias)

        def close(ours, want, scale, what):
            want = torch.as_tensor(np.asarray(want), dtype=torch.float64)
            tol = 3e-5 * max(float(scale), 1e-6)                      # fp32 convolutions in a different summation order
            err = (torch.as_tensor(ours, dtype=torch.float64) - want).abs().max().item()
            assert err <= tol, (what, case, err, tol)
        kk = c * k * k
        for name, full, pr in (("out", out.detach().cpu().double(), eps.cpu().double()), ("g_x", grads[0], probe_x.cpu().double())):
            amax = float(g[t + name + "_absmax"])
            close(full.sum((2, 3)), g[t + name + "_planes"], amax * np.sqrt(full.shape[2] * full.shape[3]), name + " plane sums")
            close(full.sum((0, 1)), g[t + name + "_pixels"], amax * np.sqrt(full.shape[0] * full.shape[1]), name + " pixel sums")
            close((full * pr).sum(), g[t + name + "_proj"], amax * np.sqrt(full.numel()), name + " projection")
            close(full[0, :, :2, :3], g[t + name + "_corner"], amax, name + " corner")
            assert abs(full.abs().max().item() - amax) <= 3e-5 * amax, name
        pw = T(conv_probe_w(seed, o, c, k)).double()
        for name, gw in (("g_wmu", grads[1]), ("g_wrho", grads[2])):
            amax = float(g[t + name + "_absmax"])
            close(gw.sum((1, 2, 3)), g[t + name + "_rowsum"], amax * np.sqrt(kk), name + " row sums")
            close(gw.sum(0).reshape(-1), g[t + name + "_colsum"], amax * np.sqrt(o), name + " column sums")
            close((gw * pw).sum(), g[t + name + "_proj"], amax * np.sqrt(o * kk), name + " projection")
            close(gw[:2, :2], g[t + name + "_corner"], amax, name + " corner")
            assert abs(gw.abs().max().item() - amax) <= 3e-5 * amax, name
        if bias:
            close(grads[3], g[t + "g_bmu"], np.abs(g[t + "g_bmu"]).max(), "g_bmu")
            close(grads[4], g[t + "g_brho"], max(np.abs(g[t + "g_brho"]).max(), 1e-3), "g_brho")
    if path == "fused":
        assert len(fused_calls) == len(g["cases"])                   # every case took the fused op


def test_svgd_fuse_auto_eligibility(backend):
    """fuse_base_optimizer="auto" fuses exactly the base optimizers whose step() the kernel reproduces."""
    ops, dev = backend

    def decide(make_base, **kw):
        torch.manual_seed(1)
        model = make_mlp().to(dev)
        ps = list(model.parameters())
        base = make_base(ps)
        opt = bde.SVGDOptimizer(ps, lambda: bde.reset_model_params(model), base, particle_count=kw.pop("m", 4),
                                dataset_size=32, _ops=ops, **kw)            # "auto" is the default
        return opt._fuse or opt._fuse_staged

    class MySGD(torch.optim.SGD):
        pass
    assert decide(lambda ps: torch.optim.SGD(ps, lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4))
    assert decide(lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-2))
    assert not decide(lambda ps: torch.optim.AdamW(ps, lr=1e-3))
    assert not decide(lambda ps: MySGD(ps, lr=0.1))
    assert not decide(lambda ps: torch.optim.Adam(ps, lr=1e-3, amsgrad=True))
    # torch >= 2.7: Adam(decoupled_weight_decay=True) is AdamW's update; the kernels apply the coupled decay
    assert not decide(lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-2, decoupled_weight_decay=True))
    assert not decide(lambda ps: torch.optim.SGD(ps, lr=0.1, maximize=True))
    assert not decide(lambda ps: torch.optim.SGD([{"params": ps[:2], "lr": 0.1}, {"params": ps[2:], "lr": 0.01}], lr=0.1))
    assert decide(lambda ps: torch.optim.SGD([{"params": ps[:2]}, {"params": ps[2:]}], lr=0.1))
    assert not decide(lambda ps: torch.optim.SGD(ps[:2], lr=0.1))                 # not the same parameter set
    assert decide(lambda ps: torch.optim.SGD(ps, lr=0.1), m=20)                  # 17..64 particles: blocked update + one apply launch

    def hooked(ps):
        base = torch.optim.SGD(ps, lr=0.1)
        base.register_step_post_hook(lambda opt, args, kwargs: None)
        return base
    assert not decide(hooked)


def test_r5_fused_svgd_refuses_late_hooks_and_diverged_groups(backend):
    """Fusability is decided at construction.  A step hook registered on the base optimizer afterwards could never run
    (the fused update does not call base.step()), and param groups whose hyper-parameters diverge cannot be applied by one
    launch: both raise a RuntimeError that says what to do, instead of silently training differently."""
    ops, dev = backend
    torch.manual_seed(1)
    x, y = torch.randn(8, 13, device=dev), torch.randn(8, 1, device=dev)

    def build():
        model = make_mlp().to(dev)
        ps = list(model.parameters())
        base = torch.optim.SGD([{"params": ps[:2]}, {"params": ps[2:]}], lr=0.05, momentum=0.9)
        opt = bde.SVGDOptimizer(ps, lambda: bde.reset_model_params(model), base, particle_count=3, dataset_size=8, _ops=ops)
        assert opt._fuse
        return model, base, opt

    def one_step(model, opt):
        return opt.step(lambda: F.mse_loss(model(x), y), lambda loss: loss.backward())
    model, base, opt = build()
    one_step(model, opt)
    base.register_step_post_hook(lambda o, a, k: None)
    with pytest.raises(RuntimeError, match="step hooks"):
        one_step(model, opt)
    model, base, opt = build()
    one_step(model, opt)
    base.param_groups[1]["lr"] = 0.5
    with pytest.raises(RuntimeError, match="identical hyper-parameters"):
        one_step(model, opt)


def test_r5_load_state_dict_warns_when_the_base_optimizer_cannot_take_over(backend):
    """After load_state_dict the optimizer the shell was constructed with takes over the loaded one's state; when it cannot
    (another optimizer class / parameter structure) the loaded, parameter-orphaned object stays in charge -- and the user is
    told, because training on with it moves nothing."""
    import copy
    import warnings
    ops, dev = backend
    torch.manual_seed(1)

    def build(make_base, n_groups=1):
        model = make_mlp().to(dev)
        ps = list(model.parameters())
        base = make_base(ps)
        return bde.SwagOptimizer(ps, base, update_interval=1, deviation_samples=3, _ops=ops), base
    src, _ = build(lambda ps: torch.optim.SGD(ps, lr=0.1, momentum=0.9))
    sd = copy.deepcopy(src.state_dict())
    same, base_same = build(lambda ps: torch.optim.SGD(ps, lr=0.3, momentum=0.9))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        same.load_state_dict(copy.deepcopy(sd))                        # same class and structure: silent hand-over
    assert same.get_base_optimizer() is base_same and base_same.param_groups[0]["lr"] == 0.1
    other, base_other = build(lambda ps: torch.optim.SGD([{"params": ps[:1]}, {"params": ps[1:]}], lr=0.1))
    with pytest.warns(RuntimeWarning, match="does not fit"):
        other.load_state_dict(copy.deepcopy(sd))
    assert other.get_base_optimizer() is not base_other


def test_bbb_group_draw_is_never_stale(backend):
    """The group-wide weight draw (rng="philox") is served only while it is current: forward passes that touch
    disjoint subsets of the group, a tensor asked twice inside one forward, a draw made under no_grad followed by a
    training forward, and parameters modified between two forwards all get fresh, differentiable samples from the
    live means (util.py:170-171 draws anew at every sample())."""
    ops, dev = backend
    torch.manual_seed(11)
    gps = [bde.GaussianParameter((6, 5), rng="philox", seed=3, _ops=ops).to(dev) for _ in range(3)]
    for gp in gps:
        gp.blundell_init()
    params = [p for gp in gps for p in gp.parameters()]
    opt = bde.BBBOptimizer(params, torch.optim.SGD(params, lr=0.5), bde.GaussianPrior(0, 1.0), dataset_size=10, _ops=ops)
    group = gps[0]._flat_group
    assert group is gps[1]._flat_group and len(group.means) == 3

    # 1. disjoint subsets: pass A touches tensors 0 and 1, pass B touches tensor 2 only, then the means move and
    #    pass C asks tensor 2 first -- it must come from the NEW means, not from the draw of pass A
    a0, a1 = gps[0].sample(), gps[1].sample()
    b2 = gps[2].sample()
    assert a0.requires_grad and b2.requires_grad
    with torch.no_grad():
        for gp in gps:
            gp.mean.add_(100.0)
    c2 = gps[2].sample()
    assert float(c2.mean()) > 90.0, "served from the draw that predates the update of the means"
    c0 = gps[0].sample()
    assert float(c0.mean()) > 90.0
    with torch.no_grad():
        for gp in gps:
            gp.mean.sub_(100.0)
    group.invalidate_draw()

    # 2. a tensor asked twice inside one forward: two DIFFERENT samples, both differentiable, and the tensors not yet
    #    served keep the group-wide draw (no second whole-group launch)
    launches = []
    real = ops.gauss_draw_fwd

    def counted(mean, rho, out, n, **k):
        launches.append(n)
        return real(mean, rho, out, n, **k)
    ops.gauss_draw_fwd = counted
    try:
        w_first, w_again = gps[0].sample(), gps[0].sample()
        w_other = gps[1].sample()
    finally:
        ops.gauss_draw_fwd = real
    assert launches == [group.gl.d, 30], launches          # one group-wide draw + one per-tensor draw of 6 x 5
    assert not torch.equal(w_first, w_again)
    (w_first.sum() + 2 * w_again.sum() + w_other.sum()).backward()
    assert torch.allclose(gps[0].mean.grad, torch.full((6, 5), 3.0, device=dev))
    assert torch.allclose(gps[1].mean.grad, torch.ones(6, 5, device=dev))
    for p in params:
        p.grad = None
    group.invalidate_draw()

    # 3. an evaluation forward under no_grad must not hand its (non-differentiable) draw to the next training forward
    with torch.no_grad():
        e0 = gps[0].sample()
    assert not e0.requires_grad
    t1 = gps[1].sample()
    assert t1.requires_grad
    t1.sum().backward()
    assert gps[1].mean.grad is not None
    for p in params:
        p.grad = None

    # 4. a step invalidates whatever was drawn before it and leaves nothing behind
    gps[0].sample()
    before = gps[2].mean.detach().clone()
    opt.step(lambda: sum(gp.sample().pow(2).sum() for gp in gps), lambda l: l.backward())
    assert group._draw is None
    assert not torch.equal(before, gps[2].mean.detach())
    after = gps[2].sample()
    assert (after.detach() - gps[2].mean.detach()).abs().max() < 1.0      # around the UPDATED mean (std ~ 0.05)

    # 5. rho modified between the forward that drew and its backward: refused (the backward re-reads rho)
    group.invalidate_draw()
    w = gps[0].sample()
    with torch.no_grad():
        gps[1].rho.add_(0.1)
    with pytest.raises(RuntimeError, match="modified in place"):
        w.sum().backward()


@pytest.mark.gpu
def test_bbb_linear_sigma_cache_follows_the_weights(backend):
    """A wide BBBLinear keeps sigma^2 of its weight matrix per weight VERSION: computed at the first forward, reused by
    the following forwards / backwards (bbb.py:63-67 runs mc_samples of them per step), recomputed after ANY in-place
    change of rho (an optimizer step, a manual edit).  Outputs and gradients equal the layer without the cache."""
    ops, dev = backend
    torch.manual_seed(17)
    prior = bde.GaussianPrior(0, 1.0)
    layers = [bde.BBBLinear(1024, 1100, prior, prior, rng="philox", seed=4, _ops=ops, sigma_cache=c).to(dev) for c in (True, False)]
    layers[1].load_state_dict(layers[0].state_dict())
    calls = []
    real = ops.lrt_sigma_cache

    def counted(*a, **k):
        calls.append(1)
        return real(*a, **k)
    ops.lrt_sigma_cache = counted
    try:
        import beyond_deep_ensembles_amd.util as U
        import itertools
        x = torch.randn(8, 1024, device=dev)

        def run(layer, stream0):
            U._philox_stream = itertools.count(stream0)
            import beyond_deep_ensembles_amd.bbb_layers as BL
            BL._philox_stream = U._philox_stream
            outs = []
            for _ in range(2):                                   # two Monte-Carlo passes on one weight version
                out = layer(x)
                grads = torch.autograd.grad(out.pow(2).sum(), [layer.weight.mean, layer.weight.rho])
                outs.append((out.detach(), grads[0], grads[1]))
            return outs
        a, b = run(layers[0], 100), run(layers[1], 100)
        assert len(calls) == 1                                    # one cache pass for two forwards + two backwards
        for (o1, g1, r1), (o2, g2, r2) in zip(a, b):
            assert torch.equal(o1, o2) and torch.equal(g1, g2) and torch.equal(r1, r2)
        with torch.no_grad():                                     # any in-place change of rho is a new version
            for layer in layers:
                layer.weight.rho.add_(0.25)
        a, b = run(layers[0], 200), run(layers[1], 200)
        assert len(calls) == 2
        for (o1, g1, r1), (o2, g2, r2) in zip(a, b):
            assert torch.equal(o1, o2) and torch.equal(g1, g2) and torch.equal(r1, r2)
        # weights edited between a forward and ITS backward: autograd refuses (the saved rho changed), cache or not
        out_old = layers[0](x)
        with torch.no_grad():
            layers[0].weight.rho.sub_(0.1)
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            torch.autograd.grad(out_old.pow(2).sum(), [layers[0].weight.mean])
        # a write through rho.data keeps address AND version counter; the layer's own invalidate call, the
        # process-wide one BBBOptimizer.step / load_state_dict issue, and nothing else, refresh the cache then
        for layer in layers:
            layer.weight.rho.data.fill_(-2.0)
        n0 = len(calls)
        layers[0].invalidate_sigma_cache()
        a, b = run(layers[0], 300), run(layers[1], 300)
        assert len(calls) == n0 + 1
        for (o1, g1, r1), (o2, g2, r2) in zip(a, b):
            assert torch.equal(o1, o2) and torch.equal(g1, g2) and torch.equal(r1, r2)
        for layer in layers:
            layer.weight.rho.data.fill_(-1.0)
        opt = bde.BBBOptimizer(layers[0].parameters(), torch.optim.SGD(layers[0].parameters(), lr=0.0), prior, dataset_size=8,
                               _ops=ops)
        opt.step(lambda: layers[0](x).pow(2).mean(), lambda l: l.backward())     # begins and ends with an invalidation
        a, b = run(layers[0], 400), run(layers[1], 400)
        for (o1, g1, r1), (o2, g2, r2) in zip(a, b):
            assert torch.equal(o1, o2) and torch.equal(g1, g2) and torch.equal(r1, r2)
    finally:
        ops.lrt_sigma_cache = real


def test_particle_set_of_the_host_helper():
    """csrc/host.cpp ParticleSet: begin()/use() re-point the parameters by moving the storage offset when they already
    view a row of the particle buffer and by set_data() otherwise (same result either way); end() hands the gradients
    over by reference and detaches them from the parameters; release() lets go of them."""
    from beyond_deep_ensembles_amd import _host
    native = _host.load()
    assert native is not None, "lib/_bde_host.so is missing or does not load: run __graft_entry__.build()"
    torch.manual_seed(5)
    shapes, m = [(3, 4), (5,), (), (2, 1, 3)], 3
    numels = [int(np.prod(s)) for s in shapes]
    offs = np.concatenate([[0], np.cumsum([(n + 3) // 4 * 4 for n in numels])]).tolist()
    ld = offs[-1]
    P, G = torch.randn(m, ld), torch.zeros(m, ld)
    params = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    views = lambda buf: [[buf[i, o:o + n].view(s) for o, n, s in zip(offs, numels, shapes)] for i in range(m)]
    pv, gv = views(P), views(G)
    ps = native.ParticleSet(params, pv, gv)
    table = torch.zeros(len(params) * m, dtype=torch.int64)
    weak = []
    for _ in range(3):
        for i in (1, 0, 2, 2, 1):                       # first call: set_data(); later ones: the offset moves
            ps.begin(i)
            for p, v, s in zip(params, pv[i], shapes):
                assert p.data_ptr() == v.data_ptr() and p.shape == torch.Size(s) and torch.equal(p.data, v) and p.grad is None
            loss = sum((p * (k + 1.0)).sum() for k, p in enumerate(params[:-1]))      # the last tensor gets no gradient
            loss.backward()
            grads = [p.grad for p in params[:-1]]
            assert ps.end(i, table, i, m, 0) == len(grads)
            assert all(p.grad is None for p in params)
            t = table.view(len(params), m)
            assert [int(t[k, i]) for k in range(len(grads))] == [g.data_ptr() for g in grads]
            assert int(t[len(params) - 1, i]) == gv[i][-1].data_ptr()                # missing: its (zeroed) flat view
            for k, g in enumerate(grads):
                assert torch.equal(g, torch.full(shapes[k], k + 1.0))
            import weakref
            weak.append(weakref.ref(grads[0]))
            del grads
        params[1].data = torch.randn(5)                 # the caller re-pointed a parameter elsewhere: set_data() again
        ps.use(2)
        assert params[1].data_ptr() == pv[2][1].data_ptr() and torch.equal(params[1].data, pv[2][1])
        ps.release()
    assert all(w() is None for w in weak)


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_svgd_many_particles_one_apply_launch_equals_the_optimizer_loop(backend, kind):
    """17..64 particles with a fusable base optimizer: blocked update + ONE bde_svgd_apply_* launch == particle_count
    calls of base.step() with its shared state (svgd.py:92-103), including what ends up in base.state."""
    ops, dev = backend

    def run(fuse):
        torch.manual_seed(11)
        model = make_mlp().to(dev)
        ps = list(model.parameters())
        base = torch.optim.SGD(ps, lr=0.05, momentum=0.9, nesterov=True, weight_decay=3e-4) if kind == "sgd" else \
            torch.optim.Adam(ps, lr=1e-3, weight_decay=1e-2)
        opt = bde.SVGDOptimizer(ps, lambda: bde.reset_model_params(model), base, particle_count=20, dataset_size=64,
                                l2_reg=0.01, fuse_base_optimizer=fuse, _ops=ops)
        assert opt._fuse_staged == fuse and not opt._fuse
        torch.manual_seed(12)
        x, y = torch.randn(16, 13, device=dev), torch.randn(16, 1, device=dev)
        losses = [float(opt.step(lambda: F.mse_loss(model(x), y), lambda l: l.backward())) for _ in range(3)]
        sd = opt.state_dict()
        return opt.particles.clone(), losses, base, ps, sd
    pa, la, base_a, ps_a, sd_a = run(True)
    pb, lb, base_b, ps_b, _ = run(False)
    np.testing.assert_allclose(pa.cpu().numpy(), pb.cpu().numpy(), rtol=3e-5, atol=3e-6)
    np.testing.assert_allclose(la, lb, rtol=1e-5)
    key = "momentum_buffer" if kind == "sgd" else "exp_avg"
    for qa, qb in zip(ps_a, ps_b):                         # the shared optimizer state, where torch keeps it
        np.testing.assert_allclose(base_a.state[qa][key].cpu().numpy(), base_b.state[qb][key].cpu().numpy(), rtol=3e-5, atol=1e-6)
    if kind == "adam":
        # (the fused paths keep ONE shared counter; the per-parameter step tensors are refreshed when the state is taken)
        assert int(base_a.state_dict()["state"][0]["step"]) == int(base_b.state_dict()["state"][0]["step"]) == 60
    assert "particle_19" in sd_a["state"][0]


def test_r6_bbb_linear_row_tiles_with_in_kernel_noise(backend):
    """BBBLinear(rng="philox", fused_linear_max_rows=...) above 128 rows: every row tile draws its activations' noise inside its
    own launch from its own Philox stream, and its backward launch REGENERATES that noise (nothing is stored).  Checked without
    knowing the streams: the noise is recovered from the output (eps = (out - mean) / sqrt(var), mean and var from the stock
    composition of bbb_layers.py:70-80), must look like a standard normal, must differ between tiles, and feeding it back as
    SUPPLIED noise must reproduce output and all five gradients."""
    ops, dev = backend
    import beyond_deep_ensembles_amd.bbb_layers as L
    torch.manual_seed(8)
    prior = bde.GaussianPrior(0, 1.0)
    b, i, o = 300, 40, 24
    layer = bde.BBBLinear(i, o, prior, prior, rng="philox", fused_linear_max_rows=512, _ops=ops).to(dev).train()
    with torch.no_grad():
        layer.weight.rho.add_(1.5)                                     # a variance large enough to recover the noise accurately
    x = torch.randn(b, i, device=dev)
    gout = torch.randn(b, o, device=dev)

    def run(noise=None):
        xin = x.clone().requires_grad_(True)
        leaves = [xin, layer.weight.mean, layer.weight.rho, layer.bias.mean, layer.bias.rho]
        real = L.normal_like
        if noise is not None:
            # supplied noise (the layer asks normal_like for it whenever the parameter does not draw inside the kernels)
            layer.weight.noise_source = lambda t: torch.zeros_like(t)
            L.normal_like = lambda t: noise.reshape(t.shape)
        try:
            out = layer(xin)
            return out.detach(), [t.detach() for t in torch.autograd.grad(out, leaves, grad_outputs=gout)]
        finally:
            layer.weight.noise_source, L.normal_like = None, real
    out, grads = run()
    with torch.no_grad():
        w, bb = layer.weight, layer.bias
        mean = F.linear(x, w.mean, bb.mean)
        var = F.linear((x ** 2).clamp(min=1e-4), (w.std ** 2).clamp(min=1e-4), (bb.std ** 2).clamp(min=1e-4))
        eps = (out - mean) / var.sqrt()
    assert abs(float(eps.mean())) < 0.05 and abs(float(eps.std()) - 1.0) < 0.05, (float(eps.mean()), float(eps.std()))
    assert not torch.allclose(eps[:128], eps[128:256], atol=1e-2)       # tiles do not share a stream
    assert not torch.allclose(eps[:44], eps[256:300], atol=1e-2)
    out2, grads2 = run(eps)
    torch.testing.assert_close(out2, out, rtol=1e-4, atol=1e-5)
    for a, c in zip(grads, grads2):
        torch.testing.assert_close(c, a, rtol=2e-3, atol=2e-4 * float(a.abs().max()))
