#!/usr/bin/env python
"""Fused SpatialTransformer head (mdx_st_head_f16) timing at a UNet level's size, cold weights (NCOPY sets cycled in one
hipGraph).  STAGES=1,2,3 gives cumulative times through the debug taps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minddiffusion_amd import ops
DEV = "cuda:0"
B, tokens = (int(v) for v in (sys.argv[1:3] if len(sys.argv) >= 3 else (2, 4096)))
C, NCOPY = 320, int(os.environ.get("NCOPY", "96"))
M = B * tokens
f16, f32 = torch.float16, torch.float32
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *s, scale=1.0, dtype=f16: (torch.randn(*s, generator=g, device=DEV) * scale).to(dtype)
x = rnd(M, C)
nrb = tokens // 128
blk = x.float().reshape(B * nrb, 128, C)
cs = torch.stack([blk.sum(1), (blk * blk).sum(1)], 2).contiguous()
packs = [ops.pack_st_head(*(rnd(C, C, scale=C ** -0.5) for _ in range(4)), *(rnd(C, scale=0.1, dtype=f32) + 1.0 for _ in range(5)))
         for _ in range(NCOPY)]
tok, qk, vt = torch.empty((M, C), dtype=f16, device=DEV), torch.empty((M, 2 * C), dtype=f16, device=DEV), torch.empty((B, C, tokens), dtype=f16, device=DEV)
dbg = torch.empty((M, C), dtype=f16, device=DEV)
for rows in (32, 64):
    for stage in [int(v) for v in os.environ.get("STAGES", "0").split(",")]:
        descs = [ops.make_st_head_desc(x, cs, nrb, s, v, tok, qk, vt, tokens, B, tokens, C, tile_rows=rows,
                                       debug_out=dbg if stage else None, debug_stage=stage) for s, v in packs]
        run = lambda: [ops.st_head_run(d) for d in descs]
        run(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            run()
        best = 1e30
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / NCOPY)
        print(f"head rows={rows} stage={stage}: {best:7.1f} us", flush=True)
