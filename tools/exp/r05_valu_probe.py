#!/usr/bin/env python
"""Issue cost of the attention softmax's VALU instructions on MI355X, fp32 forms against their half-precision counterparts (round 5,
review item 4b: would a packed-fp16 exponent path be cheaper?).  Eight independent chains per lane, one / two / three waves per SIMD."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minddiffusion_amd import _lib
lib = _lib.load()
sink = torch.zeros(4, device="cuda:0")
names = ["v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "cvt f32->f16 pair + back", "max, max, mul", "v_exp_f16", "v_pk_fma_f16", "v_pk_max_f16"]
iters = 20000
print("# tools/exp/r05_valu_probe.py (mdx_probe_valu_rate): ns per instruction (group) per wave resident on a SIMD")
for nblocks in (256, 512, 768):
    line = f"{nblocks} blocks x 256 threads ({nblocks // 256} wave(s) per SIMD):"
    for kind in range(8):
        best = 1e9
        for _ in range(3):
            _lib.check(lib.mdx_probe_valu_rate(kind, 100, nblocks, sink.data_ptr(), None), "probe")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.mdx_probe_valu_rate(kind, iters, nblocks, sink.data_ptr(), None), "probe")
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e6)
        per = best / (iters * 8) / (nblocks // 256)
        line += f"  {names[kind]} {per:.3f} ns"
    print(line, flush=True)
