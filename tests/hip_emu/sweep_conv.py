"""TEST INFRASTRUCTURE, run by hand: the fused BBBConv2d kernels (forward, input gradient, weight gradient) on the CPU model
over random layer geometries -- the bodies of test_conv_lrt_forward / test_conv_lrt_backward with CONV_CASES replaced.

    python -m tests.hip_emu.sweep_conv SEED COUNT          # random geometries (kernel 1..7, stride 1..3, padding 0..k-1)
    python -m tests.hip_emu.sweep_conv imagenet            # ResNet-18/50 layer shapes at batch 1-2
    python -m tests.hip_emu.sweep_conv batch128            # the CIFAR ResNet-20 layers at the benchmark's batch (the tilings depend on it)
    python -m tests.hip_emu.sweep_conv tilings SEED COUNT  # EVERY candidate tiling (not only the planner's choice) of random small layers
"""
import random
import sys
import time

import tests.test_ops_gpu as G
from tests.hip_emu.emu_ops import ALL, emulated

IMAGENET = [(1, 3, 64, 64, 64, 7, (2, 2), (3, 3), False), (2, 64, 56, 56, 64, 3, (1, 1), (1, 1), True),
            (1, 64, 56, 56, 128, 3, (2, 2), (1, 1), False), (1, 128, 28, 28, 128, 3, (1, 1), (1, 1), True),
            (1, 256, 14, 14, 256, 3, (1, 1), (1, 1), False), (1, 512, 7, 7, 512, 3, (1, 1), (1, 1), True),
            (1, 256, 56, 56, 64, 1, (1, 1), (0, 0), False), (1, 64, 56, 56, 256, 1, (1, 1), (0, 0), True)]


BATCH128 = [(128, 3, 32, 32, 16, 3, (1, 1), (1, 1), True), (128, 16, 32, 32, 16, 3, (1, 1), (1, 1), True),
            (128, 16, 32, 32, 32, 3, (2, 2), (1, 1), True), (128, 32, 16, 16, 32, 3, (1, 1), (1, 1), False),
            (128, 32, 16, 16, 64, 3, (2, 2), (1, 1), True), (128, 64, 8, 8, 64, 3, (1, 1), (1, 1), True),
            (128, 16, 32, 32, 32, 1, (2, 2), (0, 0), False)]


def random_cases(seed, count):
    rng = random.Random(seed)
    cases = []
    while len(cases) < count:
        k = rng.choice([1, 1, 2, 3, 3, 3, 4, 5, 7])
        n, c = rng.randint(1, 5), rng.choice([1, 2, 3, 5, 8, 16, 17, 31, 32, 33, 64, 70])
        o = rng.choice([1, 2, 7, 16, 20, 32, 33, 48, 64, 65, 100])
        h, w = rng.randint(max(k, 3), 20), rng.randint(k, 20)
        cases.append((n, c, h, w, o, k, (rng.randint(1, 3), rng.randint(1, 3)), (rng.randint(0, k - 1), rng.randint(0, k - 1)),
                      rng.random() < 0.6))
    return cases


def tilings(seed, count):
    """EVERY candidate tiling of random small layers (what tools/conv_autotune.py may pin on the device): the body of
    test_r5_conv_every_candidate_tiling_computes_the_same_layer with TILING_CASES replaced."""
    rng = random.Random(seed)
    G.DEV = "cpu"
    G.TILING_MIN_PAIRS = 0
    bad = 0
    with emulated(ALL) as ops:
        for _ in range(count):
            k = rng.choice([1, 2, 3, 3, 3, 5])
            s_ = rng.choice([1, 1, 2, 2, 3])
            case = (rng.randint(1, 3), rng.choice([2, 3, 8, 16, 17, 33, 64]), rng.randint(max(k, 4), 14), rng.randint(max(k, 4), 14),
                    rng.choice([2, 7, 16, 20, 32, 40, 64]), k, (s_, rng.choice([s_, 1])), (rng.randint(0, k - 1), rng.randint(0, k - 1)))
            n, c, h, w, o, kk, st, pd = case
            if not ops.conv_lrt_supported((n, c, h, w), (o, c, kk, kk), st, pd) or (h + 2 * pd[0] - kk) // st[0] + 1 < 1:
                continue
            G.TILING_CASES = [case]
            t = time.time()
            try:
                G.test_r5_conv_every_candidate_tiling_computes_the_same_layer(ops)
                print("ok", case, round(time.time() - t, 1), flush=True)
            except AssertionError as e:
                bad += 1
                print("FAIL", case, str(e)[:200], flush=True)
    print("failures:", bad)
    return 1 if bad else 0


def main(argv):
    if argv[0] == "tilings":
        return tilings(int(argv[1]), int(argv[2]))
    cases = IMAGENET if argv[0] == "imagenet" else BATCH128 if argv[0] == "batch128" else random_cases(int(argv[0]), int(argv[1]))
    G.DEV = "cpu"
    bad = 0
    with emulated(ALL) as ops:
        for case in cases:
            n, c, h, w, o, k, s, p, _ = case
            if not ops.conv_lrt_supported((n, c, h, w), (o, c, k, k), s, p):
                print("unsupported", case)
                continue
            G.CONV_CASES = [case]
            t = time.time()
            for fn in (G.test_conv_lrt_forward, G.test_conv_lrt_backward):
                try:
                    fn(ops)
                except AssertionError as e:
                    bad += 1
                    print("FAIL", fn.__name__, case, str(e)[:160], flush=True)
            print("ok", case, round(time.time() - t, 1), flush=True)
    print("failures:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
