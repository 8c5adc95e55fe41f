#!/usr/bin/env python
"""A/B of the eight-wave 256-pixel conv core (conv8p.hip) against the launch forms of rounds 1-3 on the conv shapes of the
batch >= 8 configurations, interleaved in one process (cdna guide 5.4 rule 24), weights rotating through cold copies.

    python tools/conv8p_bench.py [--iters 20] [--rounds 3] [--only name,...]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (name, B, H, W, Cin, Cout, skip channels)
SHAPES = [
    ("wk_l0_320_320", 16, 64, 64, 320, 320, 0),
    ("wk_l0_640_320", 16, 64, 64, 640, 320, 0),
    ("wk_l0_960_320", 16, 64, 64, 960, 320, 0),
    ("wk_l0_320_320_skip640", 16, 64, 64, 320, 320, 640),
    ("wk_l1_320_640", 16, 32, 32, 320, 640, 0),
    ("wk_l1_640_640", 16, 32, 32, 640, 640, 0),
    ("wk_l1_1280_640", 16, 32, 32, 1280, 640, 0),
    ("wk_l1_1920_640", 16, 32, 32, 1920, 640, 0),
    ("wk_l1_640_640_skip1280", 16, 32, 32, 640, 640, 1280),
    ("wk_l2_1280_1280", 16, 16, 16, 1280, 1280, 0),
    ("sd768_l0_320_320", 8, 96, 96, 320, 320, 0),
    ("sd768_l0_640_320", 8, 96, 96, 640, 320, 0),
    ("sd768_l1_640_640", 8, 48, 48, 640, 640, 0),
    ("sd768_l1_1280_640", 8, 48, 48, 1280, 640, 0),
    ("glide_l0_192_192", 16, 64, 64, 192, 192, 0),
    ("glide_l1_384_384", 16, 32, 32, 384, 384, 0),
    ("glide_sr_192_192", 8, 256, 256, 192, 192, 0),
    # "up_": nearest-2x + conv (Upsample); B, H, W = the LOW-resolution source; conv8p forms run the sub-pixel weights
    ("up_wk_32to64_640", 16, 32, 32, 640, 640, 0),
    ("up_wk_16to32_1280", 16, 16, 16, 1280, 1280, 0),
    ("up_sd768_48to96_640", 8, 48, 48, 640, 640, 0),
    ("up_b2_32to64_640", 2, 32, 32, 640, 640, 0),
    ("up_b2_16to32_1280", 2, 16, 16, 1280, 1280, 0),
    ("up_glide_32to64_384", 16, 32, 32, 384, 384, 0),
    ("b2_l0_320_320", 2, 64, 64, 320, 320, 0),
    ("b2_l1_640_640", 2, 32, 32, 640, 640, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=None)
    ap.add_argument("--bns", default="0", help="comma list of conv8p N tiles to try (0 = the library's pick)")
    ap.add_argument("--extra", default="", help="comma list of stages:tile_n forms (stages 9 = one phase per k-step)")
    ap.add_argument("--vars", default="", help="comma list of gemm_conv8p_var experiment forms to time at BN 160")
    args = ap.parse_args()
    from minddiffusion_amd import ops
    dev = torch.device("cuda:0")
    res = []
    for name, B, H, W, cin, cout, skc in SHAPES:
        if args.only and not any(o in name for o in args.only.split(",")):
            continue
        up = 1 if name.startswith("up_") else 0
        K = 9 * cin
        M = B * H * W * (4 if up else 1)
        a = torch.randn(B, H * W, cin, device=dev, dtype=torch.float16)
        xs = torch.randn(B, H * W, skc, device=dev, dtype=torch.float16) if skc else None
        wbytes = cout * K * 2
        ncopy = max(2, min(8, (300 << 20) // wbytes + 1))
        raw = [torch.randn(cout, cin, 3, 3, device=dev, dtype=torch.float16) * (K ** -0.5) for _ in range(ncopy)]
        ws = [ops.pack_conv_weight(r) for r in raw]
        wsubs = [ops.pack_subpixel_conv_weight(r) for r in raw] if up else [None] * ncopy
        del raw
        wsk = ops.pack_conv_weight(torch.randn(cout, skc, 1, 1, device=dev, dtype=torch.float16) * (skc ** -0.5)) if skc else None
        bias = torch.randn(cout, device=dev)
        emb = torch.randn(B, cout, device=dev)
        out = torch.empty(M, cout, device=dev, dtype=torch.float16)
        cst = torch.zeros(max(M // 64, 1), cout, 2, device=dev)
        skw = dict(skip_a=xs, skip_c1=skc, skip_w=wsk) if skc else {}
        forms = {"old": dict(), }
        for bn in [int(x) for x in args.bns.split(",")]:
            forms[f"c8_bn{bn}"] = dict(tile_m=256, stages=8, tile_n=bn)
        for x in [x for x in args.extra.split(",") if x]:       # e.g. "9:160" = one phase per k-step at BN 160
            st, bn = x.split(":")
            forms[f"c8_st{st}_bn{bn}"] = dict(tile_m=256, stages=int(st), tile_n=int(bn))
        for v in [int(x) for x in args.vars.split(",") if x]:
            forms[f"c8_var{v}"] = dict(tile_m=256, stages=8, tile_n=160)
        timings = {k: [] for k in forms}
        descs = {}
        wsp = None
        def route(k):       # the "old" arm is the library's choice with the eight-wave core switched off
            ops.set_option("gemm_conv8p", 0 if k == "old" else 1)
            ops.set_option("gemm_conv8p_var", int(k[6:]) if k.startswith("c8_var") else 0)
        for k, kw in forms.items():
            route(k)
            descs[k] = [ops.make_gemm_desc(a, w, cout, B, H, W, cin, out, cout, bias=bias, ksize=3, rowbias=emb, rowbias_ld=cout,
                                           colstats_out=cst, upsample=up, w_sub=None if k == "old" else wsub, **skw, **kw)
                        for w, wsub in zip(ws, wsubs)]
            need = ops.gemm_workspace_bytes(descs[k][0])
            if need and (wsp is None or wsp.numel() * 4 < need):
                wsp = ops.new_gemm_workspace(need, dev)
        for k in forms:
            for d in descs[k]:
                if wsp is not None:
                    d.workspace, d.workspace_bytes = wsp.data_ptr(), wsp.numel() * 4
        try:
            qs = {}
            for k in forms:
                route(k)
                qs[k] = ops.gemm_query(descs[k][0])
                ops.gemm_run(descs[k][0])
            torch.cuda.synchronize()
            for r in range(args.rounds):
                for k in forms:
                    route(k)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(args.iters):
                        ops.gemm_run(descs[k][i % ncopy])
                    e1.record()
                    torch.cuda.synchronize()
                    timings[k].append(e0.elapsed_time(e1) * 1e3 / args.iters)
        except Exception as e:      # a form the library refuses for this shape
            print(name, "skipped:", str(e)[:160], flush=True)
            continue
        finally:
            ops.set_option("gemm_conv8p", 1)
            ops.set_option("gemm_conv8p_var", 0)
        flops = 2.0 * M * cout * (K + skc)
        rec = dict(name=name, M=M, N=cout, K=K, skip=skc)
        line = f"{name:26s} M={M:6d} N={cout:4d} K={K:5d}"
        for k in forms:
            us = min(timings[k])
            rec[k] = dict(us=round(us, 2), tflops=round(flops / us / 1e6, 1), tile=list(qs[k][:3]),
                          all_us=[round(x, 1) for x in timings[k]])
            line += f" | {k} {qs[k][0]}x{qs[k][1]}/{qs[k][2]} {us:8.1f} us {flops / us / 1e6:7.1f} TF/s"
        res.append(rec)
        print(line, flush=True)
        del ws, a, out
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
