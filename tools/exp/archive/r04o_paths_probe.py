#!/usr/bin/env python
"""Do the LDS-DMA path and the plain-load path add up on a CU?  mdx_probe_dma_stream modes 0 (DMA), 1 (plain), 2 (both), +4 = every
block streams the same (L2-resident) region."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minddiffusion_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
buf = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)
sink = torch.zeros(4, device=dev)
def run(nblocks, waves, per, ns, mode, bpb, reps=20):
    args = (ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(bpb), nblocks, waves, per, ns, mode, 1, ctypes.c_void_p(sink.data_ptr()), None)
    _lib.check(lib.mdx_probe_dma_stream(*args), "probe"); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): lib.mdx_probe_dma_stream(*args)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best, nblocks * bpb / best / 1e3
for name, sh, bpb in (("shared 4 MiB (L2-resident)", 4, 4 << 20), ("distinct 1 MiB per block (256 MiB: MALL)", 0, 1 << 20)):
    for nblocks in (256, 512):
        for waves, per in ((4, 4), (8, 4)):
            line = f"{name} blocks={nblocks} waves={waves} per={per}:"
            for mode, mn in ((0, "dma"), (1, "plain"), (2, "both")):
                us, gbs = run(nblocks, waves, per, 3, mode + sh, bpb)
                line += f"  {mn} {us:7.1f} us {gbs/1e3:6.2f} TB/s ({gbs/256:5.1f} GB/s/CU)"
            print(line, flush=True)
