#!/usr/bin/env python
"""Per-CU / chip-wide rate of the weight path of the fused-chain kernels (coalesced 1 KiB loads to VGPRs, PF in flight)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minddiffusion_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
buf = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)
sink = torch.zeros(4, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

def run(nblocks, waves, pf, shared, bpw, reps=10):
    total = (1 if shared else nblocks) * waves * bpw + pf * 1024
    args = (ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(total), ctypes.c_uint(bpw), nblocks, waves, pf, shared,
            ctypes.c_void_p(sink.data_ptr()), st)
    _lib.check(lib.mdx_probe_l2_stream(*args), "probe")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.mdx_probe_l2_stream(*args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, nblocks * waves * bpw / us / 1e3

print("shared=1: every block streams the same waves*bpw bytes (L2-resident after the first toucher); shared=0: distinct (HBM/MALL)")
for shared in (1, 0):
    for nblocks in (128, 256):
        for waves in (4, 8, 10, 16):
            for pf in (8, 16, 32):
                bpw = 320 * 1024 if shared else 64 * 1024
                if not shared and nblocks * waves * bpw > (1 << 30):
                    continue
                us, gbs = run(nblocks, waves, pf, shared, bpw)
                print(f"shared={shared} blocks={nblocks:3d} waves={waves:2d} pf={pf:2d}: {us:8.1f} us  {gbs:8.1f} GB/s  per-CU {gbs / nblocks:6.1f} GB/s", flush=True)
