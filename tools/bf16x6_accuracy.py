#!/usr/bin/env python3
"""Numerical study for DESIGN.md section 10 item 5 (numpy, no GPU): a dot product of fp32 operands evaluated (a) as the fp32 chain
the f32 MFMA kernels run, (b) with every operand split into three bf16 terms and the six products of order <= 2^-16 accumulated
in fp32 ("bf16x6": six passes of the 16x faster bf16 matrix pipe instead of eight of the f32 pipe per 16 k), (c) with three
products only ("bf16x3"), each against fp64.  Result (200 trials, |a| ~ 0.1, |b| ~ 1):
  K = 144 :  fp32 max 1.8e-06 / mean 2.0e-07    bf16x6 2.5e-06 / 4.1e-07    bf16x3 1.5e-05 / 4.0e-06
  K = 576 :  fp32 max 3.4e-06 / mean 7.2e-07    bf16x6 1.2e-05 / 1.4e-06    bf16x3 2.8e-05 / 8.0e-06
  K = 2304:  fp32 max 1.4e-05 / mean 3.1e-06    bf16x6 2.3e-05 / 5.8e-06    bf16x3 8.5e-05 / 1.8e-05
bf16x6 with ONE accumulator is ~2x the fp32 chain's error (six times as many fp32 roundings of the accumulator): inside the
parity tests' floor (3e-6 of the output scale) but not inside "2x the reference's own fp32 deviation"; a second accumulator
for the five low-order products would bring it back to the fp32 chain's level at twice the accumulator registers."""
import numpy as np
rng=np.random.default_rng(0)
def bf16(x):  # round-to-nearest-even to bf16, returned as float32
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)
def split3(a):
    a1=bf16(a); r=(a-a1).astype(np.float32); a2=bf16(r); r2=(r-a2).astype(np.float32); a3=bf16(r2); return a1,a2,a3
for K in (144, 576, 2304):
    errs32=[];errs6=[];errs3=[]
    for trial in range(200):
        a=(rng.standard_normal(K)*0.1).astype(np.float32); b=rng.standard_normal(K).astype(np.float32)
        truth=np.dot(a.astype(np.float64),b.astype(np.float64))
        # fp32 sequential accumulate (like MFMA k-chain, fp32 products rounded? MFMA computes exact product then adds in fp32)
        acc=np.float32(0)
        for k in range(K): acc=np.float32(acc+np.float32(a[k])*np.float32(b[k]))
        errs32.append(abs(acc-truth))
        A=split3(a);B=split3(b)
        terms6=[(0,0),(0,1),(1,0),(0,2),(1,1),(2,0)]
        acc6=np.float32(0)
        # each bf16 product exact in fp32; accumulate in fp32, term by term in blocks of 16 k (like one MFMA per term)
        for k0 in range(0,K,16):
            for (i,j) in terms6[::-1]:   # small terms first
                p=(A[i][k0:k0+16].astype(np.float32)*B[j][k0:k0+16].astype(np.float32))
                for v in p: acc6=np.float32(acc6+v)
        errs6.append(abs(acc6-truth))
        acc3=np.float32(0)
        for k0 in range(0,K,16):
            for (i,j) in [(1,0),(0,1),(0,0)]:
                p=(A[i][k0:k0+16].astype(np.float32)*B[j][k0:k0+16].astype(np.float32))
                for v in p: acc3=np.float32(acc3+v)
        errs3.append(abs(acc3-truth))
    scale=np.sqrt(K)*0.1
    print(f"K={K}: fp32 chain max err {max(errs32):.2e} mean {np.mean(errs32):.2e};  bf16x6 max {max(errs6):.2e} mean {np.mean(errs6):.2e};  bf16x3 max {max(errs3):.2e} mean {np.mean(errs3):.2e}  (|sum| ~ {scale:.2f})")
