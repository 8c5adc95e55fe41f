#!/usr/bin/env python
"""Streaming-bandwidth probe: what one workgroup per CU pulls via LDS-DMA vs plain loads (see small.hip)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minddiffusion_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
buf = torch.empty(2 << 30, dtype=torch.uint8, device=dev)  # 2 GiB
buf.zero_()
sink = torch.zeros(4, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(nblocks, waves, per, ns, mode, bpb, stride=1, reps=5):
    args = (ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(bpb), nblocks, waves, per, ns, mode, stride,
            ctypes.c_void_p(sink.data_ptr()), st)
    _lib.check(lib.mdx_probe_dma_stream(*args), "probe")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.mdx_probe_dma_stream(*args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tot = nblocks * bpb
    return us, tot / us / 1e3


print("mode 0 = buffer_load..lds, mode 1 = global_load_dwordx4; GB/s total and per block")
for src_name, bpb in (("HBM (4 MiB per block, distinct)", 4 << 20), ("L2/MALL-resident (256 KiB per block, re-read)", 256 << 10)):
    for nblocks in (40, 256, 512):
        for waves, per in ((4, 8), (4, 4), (8, 4), (8, 8), (1, 8)):
            for ns in (2, 4):
                for mode in (0, 1):
                    if mode == 1 and ns != 2:
                        continue
                    if ns * waves * per * 1024 > 160 * 1024:
                        continue
                    reps = 3 if bpb > (1 << 20) else 20
                    # L2-resident case: loop the same 256 KiB 16 times by giving stride so tiles wrap (nt small) -> just repeat launches
                    us, gbs = run(nblocks, waves, per, ns, mode, bpb, 1, reps)
                    print(f"{src_name[:12]:12s} blocks={nblocks:3d} waves={waves} per={per} ns={ns} mode={mode}: {us:9.1f} us  "
                          f"{gbs:8.1f} GB/s  per-block {gbs / nblocks:6.1f} GB/s", flush=True)
