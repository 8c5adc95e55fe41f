#!/usr/bin/env python
"""GroupNorm(+SiLU) from column statistics on batch >= 8 tensors: narrow vs wide column blocks, raw vs pre-folded partials."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minddiffusion_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best
for (B, HW, C, nrb) in [(16, 4096, 320, 16), (16, 1024, 640, 4), (16, 4096, 640, 16), (8, 9216, 320, 36), (16, 1024, 1280, 4), (16, 4096, 960, 16), (2, 4096, 320, 32)]:
    xs = [torch.randn(B, HW, C, device=dev, dtype=torch.float16) for _ in range(6)]     # rotate: > L2
    out = torch.empty_like(xs[0])
    g = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
    cs = torch.randn(B * nrb, C, 2, device=dev).abs() * 100
    cs1 = torch.randn(B, C, 2, device=dev).abs() * 1600
    line = f"B={B} HW={HW} C={C} nrb={nrb} ({B*HW*C*4/1e6:.0f} MB r+w):"
    i = [0]
    def run(c, n):
        i[0] += 1
        ops.groupnorm_colstats(xs[i[0] % 6], c, n, None, None, 0, g, b, 1e-5, True, out=out)
    for name, wide, c, n in [("narrow/raw", 0, cs, nrb), ("narrow4k/raw", -1, cs, nrb), ("wide/raw", 1, cs, nrb), ("wide/folded", 1, cs1, 1)]:
        ops.set_option("gn_boost_mb", 1 if wide == -1 else 0)
        wide = max(wide, 0)
        ops.set_option("gn_wide_rows", wide)
        us = t(lambda: run(c, n))
        line += f"  {name} {us:.1f}us ({B*HW*C*4/us/1e6:.2f} TB/s)"
    ops.set_option("gn_wide_rows", 0)
    ops.set_option("gn_boost_mb", 40)
    fold = torch.empty(B, C, 2, device=dev)
    lib = __import__("minddiffusion_amd._lib", fromlist=["x"])
    us = t(lambda: lib.check(lib.load().mdx_colstats_fold_f32(cs.data_ptr(), nrb, fold.data_ptr(), 1, B, C, None), "fold"))
    line += f"  fold-launch {us:.1f}us"
    print(line, flush=True)
