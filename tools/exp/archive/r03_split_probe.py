#!/usr/bin/env python
"""Companion of r03_quant_probe.py: the same two launches (64x64 conv 320 -> 320 at UNet batch 2; GEGLU ff1 M = 512, K = 1280,
N = 10240) with forced split-K factors, hot weights and cold (512 MiB flush before every launch, HIP events around the launch)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minddiffusion_amd import ops  # noqa: E402

DEV = "cuda:0"
f16 = torch.float16
flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)


def hot(d, n=20):
    ops.gemm_run(d)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            ops.gemm_run(d)
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def cold(d, a, reps=9):
    ts = []
    for r in range(reps):
        flush.fill_(r & 1)
        a.add_(0)          # the activations are warm in a real evaluation (their producer just wrote them)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm_run(d); e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


ws = ops.new_gemm_workspace(256 << 20, DEV)
x = torch.randn(2 * 4096, 320, device=DEV).to(f16)
w = ops.pack_conv_weight((torch.randn(320, 320, 3, 3, device=DEV) * 0.02).to(f16))
out = torch.empty(2 * 4096, 320, dtype=f16, device=DEV)
print("3x3 conv 320 -> 320 at 64 x 64, UNet batch 2")
for bm, bn in ((128, 64), (128, 128), (256, 64)):
    for sk in (1, 2, 3, 5):
        d = ops.make_gemm_desc(x, w, 320, 2, 64, 64, 320, out, 320, ksize=3, tile_m=bm, tile_n=bn, splitk=sk, workspace=ws)
        try:
            q = ops.gemm_query(d)
            print(f"  tile {bm}x{bn} splitk={sk} -> {q[:3]} fixup={q[6]}: hot {hot(d):6.2f} us  cold {cold(d, x):6.2f} us", flush=True)
        except Exception as e:
            print(f"  tile {bm}x{bn} splitk={sk}: {e}")
a = torch.randn(512, 1280, device=DEV).to(f16)
w2 = ops.pack_gemm_weight((torch.randn(10240, 1280, device=DEV) * 0.02).to(f16))
out2 = torch.empty(512, 5120, dtype=f16, device=DEV)
print("GEGLU ff1 M = 512, K = 1280, N = 10240")
for bm in (128, 64):
    for sk in (1, 2, 4):
        d = ops.make_gemm_desc(a, w2, 10240, 1, 512, 1, 1280, out2, 5120, epilogue=ops.EPI_GEGLU, tile_m=bm, tile_n=128, splitk=sk,
                               workspace=ws)
        try:
            q = ops.gemm_query(d)
            print(f"  tile {bm}x128 splitk={sk} -> {q[:3]} fixup={q[6]}: hot {hot(d):6.2f} us  cold {cold(d, a):6.2f} us", flush=True)
        except Exception as e:
            print(f"  tile {bm}x128 splitk={sk}: {e}")
