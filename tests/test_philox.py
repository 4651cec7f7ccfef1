"""The in-kernel noise generator (rng="philox") is the published Philox4x32: the numpy checker reproduces the
Random123 known-answer vectors at the default 10 rounds (CPU), and the HIP kernels reproduce the checker word for word
(GPU) -- at 10 rounds (every draw of the BBB / iVON / layer kernels) and at the 7 rounds of the SWAG samplers (same
round function and key schedule, three rounds fewer)."""
import numpy as np
import pytest
import torch

from oracle import philox as PH


def test_checker_reproduces_random123_known_answers():
    for counter, key, want in PH.KAT:
        got = PH.philox4x32(np.array(counter, dtype=np.uint64), key)
        assert [int(x) for x in got] == list(want)
    # 7 rounds is a different function (guards against a silently shortened loop)
    assert [int(x) for x in PH.philox4x32(np.zeros(4, dtype=np.uint64), (0, 0), rounds=7)] != list(PH.KAT[0][2])


def test_checker_normals_are_standard():
    for rounds in (PH.ROUNDS, PH.SWAG_ROUNDS):
        z = PH.normals(1234, 0, 1 << 18, rounds=rounds)
        assert abs(z.mean()) < 6e-3 and abs(z.var() - 1) < 1e-2 and abs((z ** 4).mean() - 3) < 0.1
        assert abs(np.mean(z[1:] * z[:-1])) < 6e-3 and abs(np.mean(z[4:] * z[:-4])) < 6e-3     # neighbours / next group
    assert np.abs(z).max() < 5.78                                    # sqrt(2 * 24 ln 2): 24-bit radius uniforms
    w = PH.normals(1234, 0, 64, PH.DOMAIN_LOWRANK)
    assert not np.allclose(z[:64], w)                                # the two domains are different streams


@pytest.mark.gpu
def test_kernel_words_equal_the_known_answers_and_the_checker():
    from beyond_deep_ensembles_amd.ops import HipOps
    ops = HipOps()
    dev = "cuda:0"
    for counter, key, want in PH.KAT:
        seed = key[0] | (key[1] << 32)
        stream = counter[2] | (counter[3] << 32)
        idx0 = counter[0] | (counter[1] << 32)
        got = ops.philox_bits(seed, stream, 1, dev, domain=0, idx0=idx0).cpu().numpy()[0]
        assert [int(x) for x in got] == list(want), (counter, key)
    for seed, stream, domain, idx0, n in [(0, 0, 0, 0, 1000), (987654321987, 5, PH.DOMAIN_LOWRANK, 0, 777),
                                         (1 << 40, (1 << 33) + 7, 0, (1 << 32) - 100, 4096)]:
        for rounds in (PH.ROUNDS, PH.SWAG_ROUNDS):
            got = ops.philox_bits(seed, stream, n, dev, domain=domain, idx0=idx0, rounds=rounds).cpu().numpy()
            want = PH.stream_bits(seed, stream, n, domain, idx0, rounds=rounds).astype(np.int64)
            np.testing.assert_array_equal(got, want)
    assert ops.swag_philox_rounds == PH.SWAG_ROUNDS


@pytest.mark.gpu
def test_kernel_normals_equal_the_checker_transform():
    """Box-Muller with the bare hardware log / sqrt / sin / cos stays within 4e-6 absolute of the exact transform
    of the same words (|z| <= 5.8), and the low-rank weights use their own domain."""
    from beyond_deep_ensembles_amd.ops import HipOps
    ops = HipOps()
    dev = "cuda:0"
    d, k = 100003, 37
    e, w = torch.zeros(d, device=dev), torch.zeros(k, device=dev)
    for rounds in (PH.ROUNDS, PH.SWAG_ROUNDS):
        ops.philox_normal(4242, 11, eps_w=w, eps_d=e, rounds=rounds)
        want_e = PH.normals(4242, 11, d, rounds=rounds)
        want_w = PH.normals(4242, 11, k, PH.DOMAIN_LOWRANK, rounds=rounds)
        assert np.abs(e.cpu().numpy().astype(np.float64) - want_e).max() < 4e-6
        assert np.abs(w.cpu().numpy().astype(np.float64) - want_w).max() < 4e-6
