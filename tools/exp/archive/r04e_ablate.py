#!/usr/bin/env python
"""Timing ablations of gemm8q_kernel's main loop (option gemm_dense8q_var: results are wrong for bits 0-2, only the time counts)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minddiffusion_amd import ops
dev = torch.device("cuda:0")
for (M, N, K, epi) in [(16384, 5120, 5120, 0), (16384, 5120, 640, 1), (16384, 5120, 640, 0)]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    ws = [ops.pack_gemm_weight(torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5) for _ in range(4)]
    ncols = N // 2 if epi else N
    out = torch.empty(M, ncols, device=dev, dtype=torch.float16)
    descs = [ops.make_gemm_desc(a, w, N, 1, M, 1, K, out, ncols, epilogue=epi, tile_m=256, tile_n=256, stages=8) for w in ws]
    line = f"M={M} N={N} K={K} epi={epi}:"
    for var, name in [(0, "mfmaBurstDMA"), (32, "readBurstDMA"), (48, "readBurst+noSetprio"), (1, "noDMA"), (33, "rb:noDMA"), (4, "noMFMA"), (36, "rb:noMFMA"), (5, "noDMA+noMFMA")]:
        ops.set_option("gemm_dense8q_var", var)
        best = 1e30
        for r in range(3):
            ops.gemm_run(descs[0]); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(10):
                ops.gemm_run(descs[i % 4])
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100)
        line += f"  {name} {best:.1f}us"
    ops.set_option("gemm_dense8q_var", 0)
    print(line, flush=True)
