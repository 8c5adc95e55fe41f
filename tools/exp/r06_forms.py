#!/usr/bin/env python
"""Which kernel form every GEMM descriptor of a UNet plan resolves to (mdx_gemm_query: 0 generic, 1 HALO, 2 lean dense) + why not."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_model
from minddiffusion_amd import ops
model = sys.argv[1] if len(sys.argv) > 1 else "sd2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
h = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
net = build_model(dev, model).unet
P = net._plan(B, h, h)
cnt = collections.Counter()
for d in P.descs:
    q = ops.gemm_query(d)
    M = d.B * d.H * d.W
    key = (q[3], d.ksize)
    cnt[key] += 1
    if d.ksize == 1 and q[3] != 2:
        print("not lean: M=%d N=%d K=%d tile %dx%d split %d fixup %d epi %d out_mode %d rowbias %d out_bs %d gn %d c2 %d stride %d" % (
            M, d.N, d.c1 + d.c2, q[0], q[1], q[2], q[6], d.epilogue, d.out_mode, bool(d.rowbias), d.out_bs, bool(d.gn_colstats), d.c2, d.stride))
print(dict(cnt))
