#!/usr/bin/env python
"""Wave-quantisation probe: the 64 x 64 conv of UNet batch 2 on 128 x 64 tiles with 192 .. 512 tiles on 256 CUs (Cout varied)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minddiffusion_amd import ops
dev = torch.device("cuda:0")
B, H, W = 2, 64, 64
for cin in (320, 640):
    for cout in (192, 256, 320, 384, 448, 512, 640):
        K = 9 * cin
        a = torch.randn(B, H * W, cin, device=dev, dtype=torch.float16)
        ws = [ops.pack_gemm_weight(torch.randn(cout, K, device=dev, dtype=torch.float16) * K ** -0.5) for _ in range(8)]
        bias = torch.randn(cout, device=dev)
        out = torch.empty(B * H * W, cout, device=dev, dtype=torch.float16)
        for tm in (128, 256):
            descs = [ops.make_gemm_desc(a, w, cout, B, H, W, cin, out, cout, bias=bias, ksize=3, tile_m=tm, tile_n=64, splitk=1) for w in ws]
            wsp = ops.new_gemm_workspace(ops.gemm_workspace_bytes(descs[0]), dev)
            for d in descs:
                d.workspace, d.workspace_bytes = wsp.data_ptr(), wsp.numel() * 4
            q = ops.gemm_query(descs[0])
            best = 1e9
            for r in range(5):
                for d in descs[:2]:
                    ops.gemm_run(d)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(40):
                    ops.gemm_run(descs[i % 8])
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 40)
            tiles = (B * H * W // q[0]) * (cout // q[1])
            print(f"cin={cin} cout={cout} tile {q[0]}x{q[1]} split {q[2]} kernel {q[3]}: {tiles} tiles  {best:7.2f} us  {best / tiles * 256:6.2f} us per 256 tiles  "
                  f"{2.0 * B * H * W * cout * K / best / 1e6:6.1f} TF/s", flush=True)
