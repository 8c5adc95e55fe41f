#!/usr/bin/env python
"""Micro-benchmark of mdx_gemm_f16 on the UNet's representative conv / dense shapes.

Weights rotate through enough distinct copies to exceed the 256 MiB Infinity Cache, so the weight stream comes
from HBM as it does inside a real UNet evaluation (1.73 GB of weights per call).

    python tools/gemm_bench.py [--set b2|b16|all] [--iters 30]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (name, B, H, W, Cin, Cout, ksize, epilogue)
SHAPES_B1ROW = [
    ("conv64_320_320", 64, 64, 320, 320, 3, 0),
    ("conv64_640_320", 64, 64, 640, 320, 3, 0),
    ("conv32_640_640", 32, 32, 640, 640, 3, 0),
    ("conv32_1280_640", 32, 32, 1280, 640, 3, 0),
    ("conv16_1280_1280", 16, 16, 1280, 1280, 3, 0),
    ("conv16_2560_1280", 16, 16, 2560, 1280, 3, 0),
    ("conv8_1280_1280", 8, 8, 1280, 1280, 3, 0),
    ("conv8_2560_1280", 8, 8, 2560, 1280, 3, 0),
    ("geglu64_320", 4096, 1, 320, 2560, 1, 1),
    ("ff2_64_1280_320", 4096, 1, 1280, 320, 1, 0),
    ("qk64_320_640", 4096, 1, 320, 640, 1, 0),
    ("geglu32_640", 1024, 1, 640, 5120, 1, 1),
    ("geglu16_1280", 256, 1, 1280, 10240, 1, 1),
    ("ff2_16_5120_1280", 256, 1, 5120, 1280, 1, 0),
    ("proj16_1280", 256, 1, 1280, 1280, 1, 0),
    ("proj32_640", 1024, 1, 640, 640, 1, 0),
    ("proj8_1280", 64, 1, 1280, 1280, 1, 0),
    ("ff2_32_2560_640", 1024, 1, 2560, 640, 1, 0),
    ("qk16_1280_2560", 256, 1, 1280, 2560, 1, 0),
    ("out64_320", 4096, 1, 320, 320, 1, 0),
    ("kv64_ctx1024_320", 77, 1, 1024, 320, 1, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="2,16")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=None)
    ap.add_argument("--splits", default="0", help="comma list of split-K values to sweep (0 = auto)")
    ap.add_argument("--hot", action="store_true", help="ONE weight copy: the stream is served by the Infinity Cache, not HBM")
    args = ap.parse_args()
    from minddiffusion_amd import ops
    dev = torch.device("cuda:0")
    res = []
    for B in [int(b) for b in args.batches.split(",")]:
        for name, H, W, cin, cout, ks, epi in SHAPES_B1ROW:
            if args.only and not any(o in name for o in args.only.split(",")):
                continue
            K = ks * ks * cin
            M = B * H * W
            a = torch.randn(B, H * W, cin, device=dev, dtype=torch.float16)
            wbytes = cout * K * 2
            ncopy = 1 if args.hot else max(1, min(16, (400 << 20) // wbytes + 1))
            ws = [ops.pack_gemm_weight(torch.randn(cout, K, device=dev, dtype=torch.float16) * (K ** -0.5)) for _ in range(ncopy)]
            bias = torch.randn(cout, device=dev)
            ncols = cout // 2 if epi else cout
            out = torch.empty(M, ncols, device=dev, dtype=torch.float16)
            for sk in [int(x) for x in args.splits.split(",")]:
                if sk > 1 and K // 64 < sk:
                    continue
                descs = [ops.make_gemm_desc(a, w, cout, B, H, W, cin, out, ncols, bias=bias, ksize=ks, epilogue=epi,
                                            splitk=sk) for w in ws]
                need = ops.gemm_workspace_bytes(descs[0])
                wsp = ops.new_gemm_workspace(need, dev)
                for d in descs:
                    d.workspace = wsp.data_ptr()
                    d.workspace_bytes = wsp.numel() * 4
                for d in descs[:2]:
                    ops.gemm_run(d)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(args.iters):
                    ops.gemm_run(descs[i % ncopy])
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / args.iters
                flops = 2.0 * M * cout * K
                rec = dict(name=name, B=B, M=M, N=cout, K=K, us=round(us, 2), tflops=round(flops / us / 1e6, 1),
                           w_gbs=round(wbytes / us / 1e3, 1), split=need // max(1, M * cout * 4))
                res.append(rec)
                print(f"B={B:2d} {name:18s} M={M:6d} N={cout:5d} K={K:5d} split={rec['split']:2d} {us:9.2f} us "
                      f"{rec['tflops']:7.1f} TF/s  weights {rec['w_gbs']:7.1f} GB/s", flush=True)
            del ws, a, out
    if args.out:
        json.dump(res, open(args.out, "w"))


if __name__ == "__main__":
    main()
