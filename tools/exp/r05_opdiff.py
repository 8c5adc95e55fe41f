#!/usr/bin/env python
"""Per-shape diff of tools/op_profile.py JSONs: python tools/exp/r05_opdiff.py base.json new.json [more.json ...]"""
import collections, json, sys
runs = [json.load(open(f))["ops"] for f in sys.argv[1:]]
agg = collections.defaultdict(lambda: [0.0] * len(runs) + [0])
for ops in zip(*runs):
    k = (ops[0]["kind"], ops[0]["info"])
    for i, o in enumerate(ops):
        agg[k][i] += o["us"]
    agg[k][-1] += 1
print("per shape (sum over its ops, us): " + " | ".join(sys.argv[1:]))
for k, v in sorted(agg.items(), key=lambda kv: kv[1][-2] - kv[1][0]):
    if max(abs(x - v[0]) for x in v[:-1]) > 1.5:
        print("  " + " ".join(f"{x:8.1f}" for x in v[:-1]) + f"  x{v[-1]:2d}  {k[0]:9s} {k[1]}")
print("total   " + " ".join(f"{sum(o['us'] for o in r):8.1f}" for r in runs))
for kind in ("gemm", "groupnorm", "attention", "small"):
    print(f"{kind:8s}" + " ".join(f"{sum(o['us'] for o in r if o['kind'] == kind):8.1f}" for r in runs))
