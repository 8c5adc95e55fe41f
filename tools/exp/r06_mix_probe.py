#!/usr/bin/env python
"""Does VALU work run in the shadow of an MFMA on MI355X?  (round 6: the attention tile loop interleaves them slot by slot.)
ns per SLOT (one MFMA and / or its VALU work) per SIMD, one / two / three waves per SIMD."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minddiffusion_amd import _lib
lib = _lib.load()
sink = torch.zeros(4, device="cuda:0")
names = ["MFMA", "MFMA + 2 exp", "MFMA + 8 fma", "2 exp", "8 fma", "MFMA + pair mix", "pair mix", "MFMA(acc in AGPR) + 8 fma", "MFMA(acc AGPR) + pair mix", "MFMA(acc, A, B AGPR) + pair mix"]
iters = 20000
print("# tools/exp/r06_mix_probe.py (mdx_probe_mix_rate): ns per slot per SIMD (time / (iterations x 4 slots x waves per SIMD))")
for nblocks in (256, 512, 768):
    w = nblocks // 256
    line = f"{w} wave(s) per SIMD:"
    for kind in range(10):
        best = 1e9
        for _ in range(3):
            _lib.check(lib.mdx_probe_mix_rate(kind, 100, nblocks, sink.data_ptr(), None), "probe")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.mdx_probe_mix_rate(kind, iters, nblocks, sink.data_ptr(), None), "probe")
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e6)
        line += f"  {names[kind]} {best / (iters * 4) / w:.2f}"
    print(line, flush=True)
