#!/usr/bin/env python
"""How fast could the Upsample convs be as sub-pixel convs?  Proxy with TODAY's HALO kernel: a 3x3
conv on the LOW-resolution tensor with the same M, about the same K (4 Cin ~ 9 Cin') and N' = 4 N (the four parities as column
tiles) does the MFMA work and the operand traffic of the sub-pixel form; compared with the launch the plan issues today (generic
kernel, nearest-2x folded into the gather).  Hot weights, hipGraph of 20 launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minddiffusion_amd import ops  # noqa: E402

DEV = "cuda:0"
f16 = torch.float16


def timed(d, n=20):
    ops.gemm_run(d)
    torch.cuda.synchronize()
    g = ops.capture_graph([lambda: ops.gemm_run(d)] * n)
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


ws = ops.new_gemm_workspace(256 << 20, DEV)
for (name, B, h, cin, cout, cin_proxy) in (("32x32 -> 64x64, 640 -> 640", 2, 32, 640, 640, 320),
                                            ("16x16 -> 32x32, 1280 -> 1280", 2, 16, 1280, 1280, 576)):
    x = torch.randn(B * h * h, cin, device=DEV).to(f16)
    w = ops.pack_conv_weight((torch.randn(cout, cin, 3, 3, device=DEV) * 0.02).to(f16))
    out = torch.empty(B * 4 * h * h, cout, dtype=f16, device=DEV)
    d_now = ops.make_gemm_desc(x, w, cout, B, h, h, cin, out, cout, ksize=3, upsample=1, workspace=ws)
    t_now = timed(d_now)
    xp = torch.randn(B * h * h, cin_proxy, device=DEV).to(f16)
    wp = ops.pack_conv_weight((torch.randn(4 * cout, cin_proxy, 3, 3, device=DEV) * 0.02).to(f16))
    outp = torch.empty(B * h * h, 4 * cout, dtype=f16, device=DEV)
    d_px = ops.make_gemm_desc(xp, wp, 4 * cout, B, h, h, cin_proxy, outp, 4 * cout, ksize=3, workspace=ws)
    t_px = timed(d_px)
    print(f"{name}: today {t_now:6.1f} us (query {ops.gemm_query(d_now)[:4]}, K = {9 * cin});  sub-pixel proxy {t_px:6.1f} us "
          f"(query {ops.gemm_query(d_px)[:4]}, K = {9 * cin_proxy} vs 4 Cin = {4 * cin}, N = {4 * cout})", flush=True)
