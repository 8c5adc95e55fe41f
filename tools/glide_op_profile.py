#!/usr/bin/env python
"""Per-op HIP-event profile of the two Taichu-GLIDE UNets at the benchmarked shapes (base: 2P x 3 x 64 x 64, P = 8; up-sampler:
P x 3 x 256 x 256), ops in their real sequence, aggregated per (kind, shape) and weighted by the evaluations of one
`bench.py --config glide_256` unit (60 base + 27 up-sampler evaluations; the text prefix of a plan runs once per loop).

    python tools/glide_op_profile.py [--passes 3] [--top 40] [--out profiles/...json]

(Event pairs bracket each op, so 2-4 us of launch latency is included per op: rocprofv3 gives exact kernel times; this gives
the SHAPES behind them.)
"""
import argparse
import collections
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def profile_plan(P, passes):
    for op in P.main:
        op()
    torch.cuda.synchronize()
    best = [float("inf")] * len(P.main)
    for _ in range(passes):
        evs = []
        for op in P.main:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            op()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(evs):
            best[i] = min(best[i], a.elapsed_time(b) * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from bench import build_glide, CONFIGS
    dev = torch.device("cuda:0")
    dm, sr = build_glide(dev)
    Pn = CONFIGS["glide_256"]["batch"]
    rows = []
    for name, net, shape, evals in (("base", dm.model, (2 * Pn, 64, 64), 60), ("up", sr.model, (Pn, 256, 256), 27)):
        P = net._plan(*shape)
        P.x_static.normal_()
        P.t_static.fill_(500.0)
        P.tok_static.random_(1, 50000)
        if P.low_static is not None:
            P.low_static.normal_()
        best = profile_plan(P, args.passes)
        for i, (t, m) in enumerate(zip(best, P.meta)):
            w = 1 if m.get("text") else evals          # the text prefix runs once per loop
            rows.append(dict(model=name, idx=i, kind=m["kind"], info=m["info"], us=t, weight=w, flops=m["flops"]))
        tot = sum(r["us"] * r["weight"] for r in rows if r["model"] == name)
        print(f"{name}: {len(best)} ops, {sum(best) / 1e3:.3f} ms per evaluation (eager sum), {tot / 1e3:.1f} ms per unit", flush=True)
    unit = sum(r["us"] * r["weight"] for r in rows)
    print(f"unit (8 images): {unit / 1e3:.1f} ms summed over ops")
    kinds = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        k = kinds[(r["model"], r["kind"])]
        k[0] += r["us"] * r["weight"]
        k[1] += r["weight"]
    print("per (model, kind):")
    for k, (t, n) in sorted(kinds.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k[0]:5s} {k[1]:10s} {t / 1e3:8.1f} ms  {100 * t / unit:5.1f} %  {n:6d} launches-ish")
    shapes = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for r in rows:
        s = shapes[(r["model"], r["kind"], r["info"])]
        s[0] += r["us"] * r["weight"]
        s[1] += 1
        s[2] = max(s[2], r["us"])
    print("per shape (weighted by evaluations per unit):")
    for k, (t, n, mx) in sorted(shapes.items(), key=lambda kv: -kv[1][0])[: args.top]:
        print(f"  {k[0]:5s} {k[1]:10s} {t / 1e3:8.1f} ms  {100 * t / unit:5.1f} %  x{n:3d} per evaluation, {mx:7.1f} us worst  {k[2]}")
    if args.out:
        json.dump(dict(unit_ms=unit / 1e3, rows=rows), open(args.out, "w"))


if __name__ == "__main__":
    main()
