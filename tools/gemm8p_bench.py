#!/usr/bin/env python
"""A/B of the eight-wave 256 x 128 dense core (gemm8p.hip) against the launch forms of rounds 1-3 on the token-GEMM shapes of the
batch >= 8 configurations, interleaved in one process, weights rotating through cold copies.

    python tools/gemm8p_bench.py [--iters 20] [--rounds 3] [--only name,...]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (name, M, N, K, epilogue)   GEGLU: N = 8C
SHAPES = [
    ("wk_l1_ff1", 16384, 5120, 640, 1), ("wk_l1_ff2", 16384, 640, 2560, 0), ("wk_l1_qkv", 16384, 1920, 640, 0),
    ("wk_l1_out", 16384, 640, 640, 0),
    ("wk_l2_ff1", 4096, 10240, 1280, 1), ("wk_l2_ff2", 4096, 1280, 5120, 0), ("wk_l2_qkv", 4096, 3840, 1280, 0),
    ("wk_l2_out", 4096, 1280, 1280, 0),
    ("sd768_l1_ff1", 18432, 5120, 640, 1), ("sd768_l1_ff2", 18432, 640, 2560, 0), ("sd768_l1_out", 18432, 640, 640, 0),
    ("sd768_l2_ff1", 4608, 10240, 1280, 1), ("sd768_l2_ff2", 4608, 1280, 5120, 0), ("sd768_l2_out", 4608, 1280, 1280, 0),
    ("l0_ff1_unfused", 65536, 2560, 320, 1), ("l0_out_unfused", 65536, 320, 320, 0),
    ("b2_l1_ff1", 2048, 5120, 640, 1), ("b2_l0_ff2", 8192, 320, 1280, 0),
    # steady-state probes of the main loop (long K, plain epilogue)
    ("probe_k5120", 16384, 5120, 5120, 0), ("probe_k2560", 16384, 5120, 2560, 0), ("probe_k1280", 16384, 5120, 1280, 0),
    ("probe_k640", 16384, 5120, 640, 0), ("probe_k320", 16384, 5120, 320, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from minddiffusion_amd import ops
    dev = torch.device("cuda:0")
    res = []
    for name, M, N, K, epi in SHAPES:
        if args.only and not any(o in name for o in args.only.split(",")):
            continue
        a = torch.randn(M, K, device=dev, dtype=torch.float16)
        ncopy = max(2, min(8, (300 << 20) // (N * K * 2) + 1))
        ws = [ops.pack_gemm_weight(torch.randn(N, K, device=dev, dtype=torch.float16) * (K ** -0.5)) for _ in range(ncopy)]
        bias = torch.randn(N, device=dev)
        ncols = N // 2 if epi else N
        out = torch.empty(M, ncols, device=dev, dtype=torch.float16)
        resid = None if epi else torch.randn(M, N, device=dev, dtype=torch.float16)
        forms = {"old": dict(), "g8": dict(tile_m=256, stages=8)}
        if N % 64 == 0 and K % 64 == 0 and K >= 128:
            forms["q8"] = dict(tile_m=256, tile_n=256, stages=8)
        descs = {k: [ops.make_gemm_desc(a, w, N, 1, M, 1, K, out, ncols, bias=bias, epilogue=epi, residual=resid,
                                        residual_ld=N if resid is not None else 0, **kw) for w in ws] for k, kw in forms.items()}

        def route(k):
            ops.set_option("gemm_dense8p", 0 if k == "old" else 1)
            ops.set_option("gemm_dense8q", 0 if k == "old" else 1)
        wsp = None
        for k in forms:
            route(k)
            need = ops.gemm_workspace_bytes(descs[k][0])
            if need and (wsp is None or wsp.numel() * 4 < need):
                wsp = ops.new_gemm_workspace(need, dev)
        for k in forms:
            for d in descs[k]:
                if wsp is not None:
                    d.workspace, d.workspace_bytes = wsp.data_ptr(), wsp.numel() * 4
        timings = {k: [] for k in forms}
        qs = {}
        try:
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
        except Exception as e:
            print(name, "skipped:", str(e)[:160], flush=True)
            continue
        finally:
            ops.set_option("gemm_dense8p", 0)
            ops.set_option("gemm_dense8q", 0)
        flops = 2.0 * M * N * K
        rec = dict(name=name, M=M, N=N, K=K, epi=epi)
        line = f"{name:18s} M={M:6d} N={N:5d} K={K:5d}"
        for k in forms:
            us = min(timings[k])
            rec[k] = dict(us=round(us, 2), tflops=round(flops / us / 1e6, 1), tile=list(qs[k][:3]))
            line += f" | {k} {qs[k][0]}x{qs[k][1]}/{qs[k][2]} {us:8.1f} us {flops / us / 1e6:7.1f} TF/s"
        res.append(rec)
        print(line, flush=True)
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
