#!/usr/bin/env python3
"""Bisect of the unbatched SWAG sampler's time across builds of the library (argv: name=path ...), contiguous rows,
interleaved and repeated."""
import os, sys, subprocess, json
if len(sys.argv) > 2 or (len(sys.argv) == 2 and "=" in sys.argv[1]):
    for spec in sys.argv[1:]:
        name, path = spec.split("=")
        out = subprocess.run([sys.executable, __file__, path], capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print(f"{name:28s} {out}", flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beyond_deep_ensembles_amd import _lib
if len(sys.argv) == 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from beyond_deep_ensembles_amd.ops import HipOps
import bench
dev = torch.device("cuda", 0)
ops = HipOps()
D, K = 23_880_950, 20
ld = bench.pad_ld(D)
g = torch.Generator(device=dev).manual_seed(1)
mean = torch.randn(ld, device=dev, generator=g) * 0.05
sq = mean * mean + 1e-4
ring = torch.randn(K, ld, device=dev, generator=g) * 1e-3
theta = torch.randn(ld, device=dev, generator=g) * 0.05
o1 = torch.empty(ld, device=dev)
fn = lambda: ops.swag_sample(mean, sq, ring, 3, o1, D, seed=1, stream_id=2)
bench.time_loop(fn, 10)
ts = [bench.time_loop(fn, 20) for _ in range(5)]
tu = [bench.time_loop(lambda: ops.swag_update(theta, mean, sq, ring[3], 5, D), 20) for _ in range(3)]
print(f"swag_sample min {min(ts)*1e3:.4f} ms median {sorted(ts)[2]*1e3:.4f} ms ({4*D*(K+3)/min(ts)/8e12:.3f}); swag_update min {min(tu)*1e3:.4f} ms")
