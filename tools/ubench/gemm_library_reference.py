#!/usr/bin/env python3
"""Diagnostic (not part of the product): what the ROCm library GEMM (torch.matmul -> hipBLASLt / rocBLAS) takes for the bf16 products of
BASELINE configs[4]'s train step, timed with HIP events over back-to-back launches -- the yardstick for ppo_bf16.hpp's own GEMM kernels
(which also carry bias + tanh / TanhGrad + bias-gradient epilogues the library call does not)."""
import torch, sys
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
def t(fn, n=50, reps=10):
    """n calls captured into one graph (no host launch cost between them), replayed `reps` times"""
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(reps): g.replay()
        b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3
def rnd(*s): return torch.randn(*s, device=dev, dtype=torch.bfloat16)
print("rows per minibatch:", M)
for name, (m, k, n) in {"forward / backward hidden layer  [M x 1024] x [1024 x 1024]": (M, 1024, 1024), "first layer  [M x 256] x [256 x 1024]": (M, 256, 1024),
                        "head  [M x 1024] x [1024 x 128]": (M, 1024, 128)}.items():
    A2, B2 = rnd(2, m, k), rnd(2, k, n)
    us2 = t(lambda: torch.bmm(A2, B2))
    us1 = t(lambda: torch.matmul(A2[0], B2[0]))
    fl = 2.0 * m * k * n
    print("%-62s both towers (bmm) %7.2f us = %6.0f TFLOP/s | one tower %7.2f us = %6.0f TFLOP/s" % (name, us2, 2 * fl / us2 * 1e-6, us1, fl / us1 * 1e-6))
for name, (k, n) in {"weight gradient  [1024 x M] x [M x 1024]": (1024, 1024), "weight gradient first layer [256 x M] x [M x 1024]": (256, 1024)}.items():
    X, dY = rnd(2, M, k), rnd(2, M, n)
    us2 = t(lambda: torch.bmm(X.transpose(1, 2), dY))
    fl = 2.0 * M * k * n
    print("%-62s both towers (bmm) %7.2f us = %6.0f TFLOP/s" % (name, us2, 2 * fl / us2 * 1e-6))
