"""Seeded inputs of the BBBLinear layer fixture (tests/golden/lrt.npz): shared by oracle/gen_golden.py, which feeds them
to the reference's layer, and by the tests, which feed them to the oracle restatement and to the HIP kernels.  Test
infrastructure; imports nothing from the reference."""
import numpy as np


def lrt_case_inputs(seed, b, i, o):
    """Seeded inputs of an lrt case, regenerated identically by the tests (numpy RandomState, fp32)."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((b, i)).astype(np.float32)
    x[0, : min(i, 3)] = 0.0                                        # x^2 below the clamp
    w_mu = (rs.standard_normal((o, i)) * 0.1).astype(np.float32)
    w_rho = (rs.standard_normal((o, i)) * 1.5 - 3.0).astype(np.float32)
    w_rho[0, : min(i, 4)] = -8.0                                    # sigma^2 below the clamp
    b_mu = (rs.standard_normal(o) * 0.1).astype(np.float32)
    b_rho = (rs.standard_normal(o) - 3.0).astype(np.float32)
    eps = rs.standard_normal((b, o)).astype(np.float32)
    g = rs.standard_normal((b, o)).astype(np.float32)
    probe = rs.standard_normal((o, i)).astype(np.float32)          # projection for the [O, I] gradients
    return x, w_mu, w_rho, b_mu, b_rho, eps, g, probe


LRT_CASES = [(101, 40, 300, 70), (102, 16, 2048, 182), (103, 128, 96, 200), (104, 64, 1024, 1100), (105, 5, 13, 50)]
# batches ABOVE the fused op's 128 rows per launch (tests/golden/lrt_tiled.npz): one row past a tile, two whole tiles, seven
# tiles + a ragged eighth -- BBBLinear(fused_linear_max_rows=...) runs them as row tiles of the same kernels
LRT_TILED_CASES = [(201, 129, 96, 70), (202, 256, 300, 50), (203, 1000, 48, 24)]
