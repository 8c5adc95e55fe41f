#!/usr/bin/env python
"""Fused SpatialTransformer tail (mdx_st_tail_f16) vs the unfused launches it replaces, at a UNet level's real size, with COLD
weights: NCOPY distinct weight sets are cycled (NCOPY x 3.3 MB > the 256 MB Infinity Cache), the whole cycle is one hipGraph,
time = graph time / NCOPY.  Usage: stchain_bench.py [B tokens heads]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from minddiffusion_amd import ops  # noqa: E402

DEV = "cuda:0"
B, tokens, heads = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (2, 4096, 5)))
C, ctx_len, ctx_cap = 320, 77, 80
NCOPY = int(os.environ.get("NCOPY", "96"))
REPS = int(os.environ.get("REPS", "1"))     # passes over the weight sets per graph (NCOPY=1 REPS=50: weights hot in L2)
M = B * tokens
d = C // heads
f16, f32 = torch.float16, torch.float32
g = torch.Generator(device=DEV).manual_seed(0)


def rnd(*shape, scale=1.0, dtype=f16):
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(dtype)


x = dict(attn_o=rnd(M, C), tok=rnd(M, C), x_in=rnd(M, C), k=rnd(B, ctx_cap, C), vt=rnd(B, C, ctx_cap))
sets = []
for _ in range(NCOPY):
    w = {n: rnd(*s, scale=s[1] ** -0.5) for n, s in (("o1", (C, C)), ("q2", (C, C)), ("o2", (C, C)), ("ff1", (8 * C, C)),
                                                      ("ff2", (C, 4 * C)), ("po", (C, C)))}
    for n, k in (("bo1", C), ("bo2", C), ("b1", 8 * C), ("b2", C), ("bpo", C), ("be2", C), ("be3", C), ("g2", C), ("g3", C)):
        w[n] = rnd(k, scale=0.1, dtype=f32) + (1.0 if n[0] == "g" else 0.0)
    sets.append(w)


def timed_graph(build):
    """build() enqueues one pass over all weight sets; returns us per set (graph replay, min of 5)."""
    build()
    torch.cuda.synchronize()
    if os.environ.get("NOGRAPH"):       # eager launches (rocprofv3 --pmc cannot follow graph replays)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            build()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (NCOPY * REPS)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(REPS):
            build()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (NCOPY * REPS))
    return best


# ---- fused
res = {}
STAGES = [int(v) for v in os.environ.get("STAGES", "0").split(",")]    # debug taps: cumulative time up to a stage
dbg = torch.empty((M, C), dtype=f16, device=DEV)
for tile_rows, stage in [(r, s) for r in (64, 32) for s in STAGES]:
    packs = [ops.pack_st_tail(*(w[n] for n in ("o1", "q2", "o2", "ff1", "ff2", "po", "bo1", "g2", "be2", "bo2", "g3", "be3", "b1",
                                               "b2", "bpo"))) for w in sets]
    out = torch.empty((M, C), dtype=f16, device=DEV)
    cs = torch.zeros((M // tile_rows, C, 2), dtype=f32, device=DEV)
    descs = [ops.make_st_tail_desc(x["attn_o"], x["tok"], x["x_in"], out, x["k"], x["vt"], s, v, B, tokens, C, heads, d, ctx_len,
                                   ctx_cap, tile_rows=tile_rows, colstats_out=cs, debug_out=dbg if stage else None, debug_stage=stage)
             for s, v in packs]
    us = timed_graph(lambda: [ops.st_tail_run(dd) for dd in descs])
    if stage:
        print(f"fused tile_rows={tile_rows} up to stage {stage}: {us:8.1f} us", flush=True)
        continue
    res[f"fused_r{tile_rows}"] = us
    fl = 2.0 * M * 16 * C * C + 4.0 * M * ctx_len * C
    print(f"fused tile_rows={tile_rows}: {us:8.1f} us per block  ({fl / us / 1e6:6.1f} TF/s)", flush=True)

if os.environ.get("FUSED_ONLY"):
    sys.exit(0)
# ---- unfused (what the UNet plan launches today)
ws = ops.new_gemm_workspace(64 << 20, DEV)
st = torch.zeros((M, C // 64, 2), dtype=f32, device=DEV)
bufs = dict(t1=torch.empty((M, C), dtype=f16, device=DEV), q2=torch.empty((M, C), dtype=f16, device=DEV),
            o2=torch.empty((M, C), dtype=f16, device=DEV), t2=torch.empty((M, C), dtype=f16, device=DEV),
            g=torch.empty((M, 4 * C), dtype=f16, device=DEV), t3=torch.empty((M, C), dtype=f16, device=DEV),
            out=torch.empty((M, C), dtype=f16, device=DEV))
keep = []
chains = []
for w in sets:
    pk = lambda t: ops.pack_gemm_weight(t)
    wq, sq, cbq = ops.fold_layernorm(w["q2"], w["g2"], w["be2"])
    half, nt = 4 * C, 4 * C // 64
    b1i = torch.stack([w["b1"][:half].reshape(nt, 64), w["b1"][half:].reshape(nt, 64)], 1).reshape(-1).contiguous()
    w1i = torch.stack([w["ff1"][:half].reshape(nt, 64, C), w["ff1"][half:].reshape(nt, 64, C)], 1).reshape(2 * half, C)
    w1f, s1, cb1 = ops.fold_layernorm(w1i, w["g3"], w["be3"], b1i)
    P = dict(o1=pk(w["o1"]), q2=pk(wq), o2=pk(w["o2"]), ff1=pk(w1f), ff2=pk(w["ff2"]), po=pk(w["po"]))
    keep.append((P, sq, cbq, s1, cb1))
    mk = lambda a, wt, N, K, out, **kw: ops.make_gemm_desc(a, wt, N, B, tokens, 1, K, out, out.shape[1], workspace=ws, **kw)
    ds = [mk(x["attn_o"], P["o1"], C, C, bufs["t1"], bias=w["bo1"], residual=x["tok"], residual_ld=C, stats_out=st),
          mk(bufs["t1"], P["q2"], C, C, bufs["q2"], bias=cbq, ln_stats=st, ln_s=sq),
          None,
          mk(bufs["o2"], P["o2"], C, C, bufs["t2"], bias=w["bo2"], residual=bufs["t1"], residual_ld=C, stats_out=st),
          mk(bufs["t2"], P["ff1"], 8 * C, C, bufs["g"], bias=cb1, ln_stats=st, ln_s=s1, epilogue=ops.EPI_GEGLU),
          mk(bufs["g"], P["ff2"], C, 4 * C, bufs["t3"], bias=w["b2"], residual=bufs["t2"], residual_ld=C),
          mk(bufs["t3"], P["po"], C, C, bufs["out"], bias=w["bpo"], residual=x["x_in"], residual_ld=C)]
    chains.append(ds)


def run_unfused():
    for ds in chains:
        for dd in ds:
            if dd is None:
                ops.attention(bufs["q2"].data_ptr(), x["k"].data_ptr(), x["vt"].data_ptr(), bufs["o2"].data_ptr(), B, heads, d,
                              tokens, ctx_len, d ** -0.5, tokens * C, C, ctx_cap * C, C, C * ctx_cap, ctx_cap, tokens * C, C)
            else:
                ops.gemm_run(dd)


us = timed_graph(run_unfused)
print(f"unfused chain (7 launches): {us:8.1f} us per block", flush=True)
for k, v in res.items():
    print(f"  {k}: {v:.1f} us = {us / v:.2f}x")
