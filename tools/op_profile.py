#!/usr/bin/env python
"""Per-op HIP-event profile of one UNet evaluation (eager, ops in their real sequence).

    python tools/op_profile.py --batch 2 --latent 64 [--passes 5] [--top 30]

Prints time per op kind and the slowest individual ops with their shapes; writes JSON to --out.
(Event pairs bracket each op, so ~2-4 us of launch latency is included per op: use rocprofv3 for exact kernel times.)
"""
import argparse
import collections
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--model", default="sd2")
    ap.add_argument("--out", default=None)
    ap.add_argument("--guidance", action="store_true", help="profile the op list of a guidance batch (cat([x] * 2), UNetModel._dup_body: "
                    "conv_in .. attn1.to_out of the first SpatialTransformer from the half-batch plan); needs batch >= ops option unet_cfg_dup")
    args = ap.parse_args()
    from bench import build_model, build_vae
    dev = torch.device("cuda:0")
    B, h = args.batch, args.latent
    if args.model == "vae":     # AutoencoderKL.decode of the SD VAE
        dec = build_vae(dev).decoder
        P = dec._plan(B, h, h)
        P.z_static.copy_(torch.randn(B, 4, h, h, device=dev))
    else:
        model = build_model(dev, args.model)
        net = model.unet
        P = net._plan(B, h, h)
        ctx = torch.randn(B, 77, net.context_dim, device=dev, dtype=torch.float16)
        net._ensure_context(P, ctx)
        P.x_static.copy_(torch.randn(B, 4, h, h, device=dev))
        P.t_static.fill_(501.0)
    oplist, metas = P.main, P.meta
    if args.guidance:
        for op in P.main[:P.temb_ops]:
            op()
        body = net._dup_body(P)
        assert body is not None, "no guidance-duplicate prefix for this batch / network"
        oplist, metas = body, P.dup_meta
    for op in oplist:
        op()
    torch.cuda.synchronize()
    best = [float("inf")] * len(oplist)
    for _ in range(args.passes):
        evs = []
        for op in oplist:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            op()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(evs):
            best[i] = min(best[i], a.elapsed_time(b) * 1e3)  # us
    kinds = collections.defaultdict(lambda: [0.0, 0, 0])
    for t, m in zip(best, metas):
        k = kinds[m["kind"]]
        k[0] += t
        k[1] += 1
        k[2] += m["flops"]
    total = sum(best)
    print(f"{args.model} eval B={B} latent={h}: sum of per-op times {total / 1e3:.3f} ms over {len(best)} ops")
    for k, (t, n, f) in sorted(kinds.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:10s} {t / 1e3:8.3f} ms  {n:4d} ops  {f / 1e12:7.4f} TFLOP  {f / max(t, 1e-9) / 1e6:8.1f} TFLOP/s")
    order = sorted(range(len(best)), key=lambda i: -best[i])[: args.top]
    print("slowest ops:")
    for i in order:
        m = metas[i]
        print(f"  #{i:3d} {m['kind']:10s} {best[i]:8.1f} us  {m['flops'] / max(best[i], 1e-9) / 1e6:7.1f} TF/s  {m['info']}")
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"B": B, "latent": h, "total_us": total,
                       "ops": [dict({k: v for k, v in m.items() if k != "desc"}, us=t) for t, m in zip(best, metas)]}, f)


if __name__ == "__main__":
    main()
