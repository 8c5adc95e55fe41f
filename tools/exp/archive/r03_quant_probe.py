#!/usr/bin/env python
"""Does a 320-tile launch on 256 CUs pay for two rounds?  Times the 64x64-level 3x3 conv (320 -> N, UNet batch 2: 64 M tiles of
128 pixels x N / 64 column tiles) and the 16x16-level GEGLU ff1 (M = 512, K = 1280, N columns in 128 x 128 tiles) for N chosen so
that the grid has 192 / 256 / 320 / 384 / 512 tiles, weights hot (one copy, hipGraph of 20 launches): if time followed the work it
would be proportional to the tile count; a staircase at 256 says the excess tiles cost a whole round."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minddiffusion_amd import ops  # noqa: E402

DEV = "cuda:0"
f16, f32 = torch.float16, torch.float32


def timed(d, n=20):
    ops.gemm_run(d)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            ops.gemm_run(d)
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


ws = ops.new_gemm_workspace(64 << 20, DEV)
print("3x3 conv 320 -> N at 64 x 64, UNet batch 2, 128 x 64 HALO tiles")
x = torch.randn(2 * 4096, 320, device=DEV).to(f16)
for N in (192, 256, 320, 384, 512):
    w = ops.pack_conv_weight((torch.randn(N, 320, 3, 3, device=DEV) * 0.02).to(f16))
    out = torch.empty(2 * 4096, N, dtype=f16, device=DEV)
    d = ops.make_gemm_desc(x, w, N, 2, 64, 64, 320, out, N, ksize=3, tile_m=128, tile_n=64, splitk=1, workspace=ws)
    q = ops.gemm_query(d)
    us = timed(d)
    tiles = 64 * (N // 64)
    print(f"  N={N:4d} tiles={tiles:4d} query={q[:4]} {us:7.2f} us  {us / tiles * 256:7.2f} us per 256 tiles  "
          f"{2 * 8192 * N * 2880 / us / 1e6:6.1f} TF/s", flush=True)
print("GEGLU ff1 M = 512, K = 1280, 128 x 128 tiles")
a = torch.randn(512, 1280, device=DEV).to(f16)
for N in (6144, 8192, 10240, 12288, 16384):
    w = ops.pack_gemm_weight((torch.randn(N, 1280, device=DEV) * 0.02).to(f16))
    out = torch.empty(512, N // 2, dtype=f16, device=DEV)
    d = ops.make_gemm_desc(a, w, N, 1, 512, 1, 1280, out, N // 2, epilogue=ops.EPI_GEGLU, tile_m=128, tile_n=128, splitk=1,
                           workspace=ws)
    q = ops.gemm_query(d)
    us = timed(d)
    tiles = 4 * (N // 128)
    print(f"  N={N:5d} tiles={tiles:4d} query={q[:4]} {us:7.2f} us  {us / tiles * 256:7.2f} us per 256 tiles  "
          f"{2 * 512 * N * 1280 / us / 1e6:6.1f} TF/s", flush=True)
