"""Philox4x32-10 in numpy -- the checker for the in-kernel noise generator.

TEST INFRASTRUCTURE ONLY (see oracle/bde_oracle.py): nothing under
``beyond_deep_ensembles_amd/`` imports this.

The reference draws its noise with ``torch.randn`` (``util.py:185-186``,
``swag.py:57`` through ``LowRankMultivariateNormal.rsample``, ``ivorn.py:108``);
the HIP path offers that stream (``rng="torch"``, noise supplied by the caller)
AND an in-kernel counter-based generator (``rng="philox"``) whose normals never
touch HBM.  The generator is the published Philox4x32-10 of Salmon, Moraes,
Dror & Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11); the
algorithm below is restated from the paper, and pinned by the known-answer
vectors of the Random123 distribution (``KAT`` below), which the GPU kernel
must reproduce word for word through ``bde_philox_bits``.

Counter layout used by the kernels (csrc/bde_common.hpp): counter =
``(lo32(g), hi32(g), lo32(stream), hi32(stream) ^ domain)`` for float4 group
``g``, key = ``(lo32(seed), hi32(seed))``; element ``4 g + j`` of a stream gets
normal ``j`` of the Box-Muller pair transform of the four output words.
"""
from __future__ import annotations

import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57          # round multipliers
W0, W1 = 0x9E3779B9, 0xBB67AE85          # Weyl key increments (golden ratio, sqrt(3) - 1)
DOMAIN_DIAG, DOMAIN_LOWRANK = 0x0, 0x80000000
ROUNDS = 10            # the published default: every draw of the BBB / iVON / layer kernels
SWAG_ROUNDS = 7        # the SWAG samplers' noise (bde_swag_philox_rounds): the fewest rounds the paper reports as
                       # Crush-resistant; the same round function and key schedule, three rounds fewer

# (counter[4], key[2]) -> output[4]; Random123 kat_vectors, philox4x32 with 10 rounds
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF),
     (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def philox4x32(counter: np.ndarray, key, rounds: int = 10) -> np.ndarray:
    """``counter [..., 4]`` uint32 words, ``key`` = two uint32 words -> ``[..., 4]`` uint32."""
    c = np.asarray(counter, dtype=np.uint64) & 0xFFFFFFFF
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(rounds):
        p0, p1 = c0 * M0, c2 * M1                       # 32 x 32 -> 64 bit products
        hi0, lo0 = p0 >> 32, p0 & 0xFFFFFFFF
        hi1, lo1 = p1 >> 32, p1 & 0xFFFFFFFF
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def stream_bits(seed: int, stream_id: int, n_groups: int, domain: int = DOMAIN_DIAG, idx0: int = 0,
                rounds: int = ROUNDS) -> np.ndarray:
    """The words ``bde_philox_bits(seed, stream_id, domain, idx0, ..., n_groups, rounds)`` must return: ``[n_groups, 4]``."""
    g = (np.arange(n_groups, dtype=np.uint64) + np.uint64(idx0 & 0xFFFFFFFFFFFFFFFF))
    ctr = np.empty((n_groups, 4), dtype=np.uint64)
    ctr[:, 0] = g & np.uint64(0xFFFFFFFF)
    ctr[:, 1] = g >> np.uint64(32)
    ctr[:, 2] = stream_id & 0xFFFFFFFF
    ctr[:, 3] = ((stream_id >> 32) & 0xFFFFFFFF) ^ domain
    return philox4x32(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF), rounds)


def box_muller(bits: np.ndarray) -> np.ndarray:
    """``[n, 4]`` words -> ``[n, 4]`` float64 standard normals, the kernels' transform evaluated exactly:
    u = (top 24 bits + 1) / 2^24 for the radius words (0 and 2), top 24 bits / 2^24 (in revolutions) for the
    angle words (1 and 3); (z0, z1) = r0 (cos, sin)(2 pi u1), (z2, z3) = r1 (cos, sin)(2 pi u3)."""
    b = np.asarray(bits, dtype=np.uint64)
    u0 = ((b[:, 0] >> 8).astype(np.float64) + 1.0) / 16777216.0
    u1 = (b[:, 1] >> 8).astype(np.float64) / 16777216.0
    u2 = ((b[:, 2] >> 8).astype(np.float64) + 1.0) / 16777216.0
    u3 = (b[:, 3] >> 8).astype(np.float64) / 16777216.0
    r0, r1 = np.sqrt(-2.0 * np.log(u0)), np.sqrt(-2.0 * np.log(u2))
    a0, a1 = 2.0 * np.pi * u1, 2.0 * np.pi * u3
    return np.stack([r0 * np.cos(a0), r0 * np.sin(a0), r1 * np.cos(a1), r1 * np.sin(a1)], axis=1)


def normals(seed: int, stream_id: int, n: int, domain: int = DOMAIN_DIAG, rounds: int = ROUNDS) -> np.ndarray:
    """float64 normals of elements ``0 .. n-1`` of a stream (what ``bde_philox_normal`` writes, before fp32 rounding)."""
    groups = (n + 3) // 4
    return box_muller(stream_bits(seed, stream_id, groups, domain, rounds=rounds)).reshape(-1)[:n]
