#!/usr/bin/env python
"""Micro-benchmark of mdx_attention_f16 at the UNet's self-attention shapes.

    python tools/attn_bench.py [--shapes B,heads,N,D;...] [--iters 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="8,5,9216,64;2,5,4096,64;16,8,4096,40;16,10,1024,64")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--forms", default="o2,o3,o3s,p",
                    help="launch forms to time, interleaved: o2 / o3 = four-wave kernel built for two / three blocks per CU; a trailing "
                         "'s' adds the split-KV workspace with the library's own split choice, 'sN' forces N splits; a8 = the eight-wave kernel")
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    from minddiffusion_amd import ops
    dev = torch.device("cuda:0")
    for spec in args.shapes.split(";"):
        B, h, N, D = (int(v) for v in spec.split(","))
        inner = h * D
        qk = torch.randn(B, N, 2 * inner, device=dev, dtype=torch.float16)        # [q | k] as the merged projection writes them
        vt = torch.randn(B, inner, N, device=dev, dtype=torch.float16)            # V^T
        o = torch.empty(B, N, inner, device=dev, dtype=torch.float16)
        items = (N + 127) // 128 * h * B
        ws = ops.attention_workspace(65536 + items * 8 * (128 * D * 2 + 1024), dev)

        def setup(form):
            ops.set_option("attn8", 2 if form == "a8" else 0)
            ops.set_option("attn_pipe", 1 if form.startswith("p") else 0)      # p = the software-pipelined kernel (round 6)
            ops.set_option("attn_occ3", 0 if form.startswith("o2") else 1)
            if "s" in form:
                n = form.split("s")[1]
                return ws, (int(n) if n else 0)
            return None, 0

        def run(w, ns):
            ops.attention(qk.data_ptr(), qk.data_ptr() + inner * 2, vt.data_ptr(), o.data_ptr(), B, h, D, N, N, D ** -0.5,
                          N * 2 * inner, 2 * inner, N * 2 * inner, 2 * inner, inner * N, N, N * inner, inner, ws=w, kv_splits=ns)
        forms = args.forms.split(",")
        pipe_default = ops.get_option("attn_pipe")
        best = {f: 1e30 for f in forms}
        allt = {f: [] for f in forms}
        try:
            for _ in range(args.rounds):
                for f in forms:
                    w, ns = setup(f)
                    run(w, ns)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        run(w, ns)
                    e1.record()
                    torch.cuda.synchronize()
                    best[f] = min(best[f], e0.elapsed_time(e1) * 1e3 / args.iters)
                    allt[f].append(e0.elapsed_time(e1) * 1e3 / args.iters)
        finally:
            ops.set_option("attn8", 0)
            ops.set_option("attn_occ3", 1)
            ops.set_option("attn_pipe", pipe_default)
        line = f"self-attention B={B} heads={h} N={N} D={D}:"
        for f in forms:
            med = sorted(allt[f])[len(allt[f]) // 2]
            line += f"  {f} {best[f]:8.1f} us (median {med:.1f}) {4.0 * B * h * N * N * D / best[f] / 1e6:6.1f} TF/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
