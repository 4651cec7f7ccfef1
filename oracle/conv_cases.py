"""Seeded inputs of the BBBConv2d layer fixture (tests/golden/conv_lrt.npz): shared by oracle/gen_golden.py, which feeds
them to the reference's layer, and by the tests, which feed them to this package's layer (CPU checker and HIP kernels).
Test infrastructure; imports nothing from the reference."""
import numpy as np


def conv_case_inputs(seed, n, c, h, w, o, k, stride, padding):
    """Seeded inputs of a conv case, regenerated identically by the tests (numpy RandomState, fp32)."""
    rs = np.random.RandomState(seed)
    ho, wo = (h + 2 * padding - k) // stride + 1, (w + 2 * padding - k) // stride + 1
    x = rs.standard_normal((n, c, h, w)).astype(np.float32)
    x[0, 0, :2] = 0.0                                              # x^2 below the clamp (exact zeros)
    x[0, 0, 2, : min(w, 3)] = 5e-3                                 # ... and below it without being zero
    w_mu = (rs.standard_normal((o, c, k, k)) * 0.1).astype(np.float32)
    w_rho = (rs.standard_normal((o, c, k, k)) * 1.5 - 3.0).astype(np.float32)
    w_rho[0, 0] = -8.0                                             # sigma^2 below the clamp
    b_mu = (rs.standard_normal(o) * 0.1).astype(np.float32)
    b_rho = (rs.standard_normal(o) - 3.0).astype(np.float32)
    eps = rs.standard_normal((n, o, ho, wo)).astype(np.float32)
    g = rs.standard_normal((n, o, ho, wo)).astype(np.float32)
    probe_x = rs.standard_normal((n, c, h, w)).astype(np.float32)  # projections for the large gradients
    return x, w_mu, w_rho, b_mu, b_rho, eps, g, probe_x


def conv_probe_w(seed, o, c, k):
    """Seeded projection tensor for the [O, C, K, K] gradients."""
    return np.random.RandomState(seed + 7919).standard_normal((o, c, k, k)).astype(np.float32)


# (seed, N, C, H, W, O, K, stride, padding, bias): the CIFAR ResNet-20 layer shapes of BASELINE configs[1] at a small batch
# (the first layer, the three stages, both stride-2 transitions, a 1x1 stride-2 shortcut) and two ragged geometries
CONV_CASES = [(201, 4, 3, 32, 32, 16, 3, 1, 1, 1), (202, 4, 16, 32, 32, 16, 3, 1, 1, 1), (203, 4, 16, 32, 32, 32, 3, 2, 1, 1),
              (204, 4, 32, 16, 16, 32, 3, 1, 1, 0), (205, 4, 32, 16, 16, 64, 3, 2, 1, 1), (206, 6, 64, 8, 8, 64, 3, 1, 1, 1),
              (207, 4, 16, 32, 32, 32, 1, 2, 0, 0), (208, 3, 5, 9, 11, 7, 3, 1, 0, 1), (209, 2, 7, 13, 6, 33, 5, 1, 2, 1)]
