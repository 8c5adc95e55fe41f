#!/usr/bin/env python
"""Per-block phase timeline of one mdx_gemm_f16 launch (mdx_probe_gemm_trace): where inside the kernel the time goes.

    python tools/gemm_trace.py --shape M,N,K[,ksize,H,W] [--split S] [--batch B]

Phases (100 MHz realtime counter, shown in us relative to the earliest block start):
  start -> prologue DMAs issued -> first K tile landed -> main loop done -> epilogue done.
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the trace marks exist only in the diagnostics build (make -C minddiffusion_amd/csrc trace)
os.environ.setdefault("MDX_LIBRARY", os.path.join(ROOT, "minddiffusion_amd", "libmdx_trace.so"))
if not os.path.exists(os.environ["MDX_LIBRARY"]):      # the diagnostics build does not travel to the GPU box (.gpurunignore): build it there
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "minddiffusion_amd", "csrc"), "-j16", "trace"], check=True,
                   stdout=subprocess.DEVNULL)


def trace_one(ops, lib, name, B, H, W, cin, cout, ks, sk, reps=3, flush_caches=True):
    dev = torch.device("cuda:0")
    K = ks * ks * cin
    M = B * H * W
    a = torch.randn(B, H * W, cin, device=dev, dtype=torch.float16)
    out = torch.empty(M, cout, device=dev, dtype=torch.float16)
    bias = torch.randn(cout, device=dev)
    tbuf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    res = []
    for r in range(reps):
        w = ops.pack_gemm_weight(torch.randn(cout, K, device=dev, dtype=torch.float16) * (K ** -0.5))  # cold weights
        d = ops.make_gemm_desc(a, w, cout, B, H, W, cin, out, cout, bias=bias, ksize=ks, splitk=sk)
        need = ops.gemm_workspace_bytes(d)
        wsp = ops.new_gemm_workspace(need, dev)
        d.workspace, d.workspace_bytes = wsp.data_ptr(), wsp.numel() * 4
        flush = None
        if flush_caches:
            flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev).fill_(r)   # push everything out of L2/MALL
        else:
            ops.gemm_run(d)                                                            # warm run: operands + output resident
        torch.cuda.synchronize()
        tbuf.zero_()
        lib.mdx_probe_gemm_trace(ctypes.c_void_p(tbuf.data_ptr()), ctypes.c_size_t(tbuf.numel() * 8))
        ops.gemm_run(d)
        lib.mdx_probe_gemm_trace(None, 0)
        torch.cuda.synchronize()
        t = tbuf.view(-1, 8).cpu()
        t5 = t[t[:, 0] != 0][:, 5].double() / 100.0
        t = t[t[:, 0] != 0][:, :5].double() / 100.0   # us
        t0 = t[:, 0].min()
        t = t - t0
        stg = (t5 - t0 - t[:, 3]) if float(t5.max()) > 0 else None    # epilogue: accumulators staged in LDS (slot 5)
        res.append(t)
        del flush
    t = res[-1]
    nb = t.shape[0]
    names = ["start", "prologue issued", "first tile landed", "main loop done", "epilogue done"]
    print(f"{name}: M={M} N={cout} K={K} split={sk} -> {nb} blocks; kernel span {t[:, 4].max():.2f} us")
    for i, nm in enumerate(names):
        c = t[:, i]
        print(f"   {nm:18s} mean {c.mean():7.2f}  min {c.min():7.2f}  max {c.max():7.2f}")
    d = t[:, 1:] - t[:, :-1]
    print("   per-block phase durations (mean us): setup %.2f | first-tile wait %.2f | main loop %.2f | epilogue %.2f"
          % tuple(d.mean(0).tolist()) + ("" if stg is None else "  (of which LDS staging + barrier %.2f)" % float(stg.mean())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--split", type=int, default=0)
    ap.add_argument("--warm", action="store_true", help="do not flush the caches: operands and output stay resident")
    ap.add_argument("--only", default="", help="comma list of shape-name substrings")
    ap.add_argument("--lean", type=int, default=-1, help="0 / 1: library option gemm_lean_dense (round 6: csrc/dense.hip); -1 = default")
    args = ap.parse_args()
    from minddiffusion_amd import ops, _lib
    lib = _lib.load()
    if args.lean >= 0:
        ops.set_option("gemm_lean_dense", args.lean)
    print(f"# gemm_lean_dense = {ops.get_option('gemm_lean_dense')}")
    B = args.batch
    shapes = [("proj16_1280", 16, 16, 1280, 1280, 1), ("proj32_640", 32, 32, 640, 640, 1),
              ("qk64_320", 64, 64, 320, 320, 1), ("conv16_1280_1280", 16, 16, 1280, 1280, 3),
              ("conv32_640_640", 32, 32, 640, 640, 3), ("conv64_320_320", 64, 64, 320, 320, 3),
              ("ff2_16_5120_1280", 16, 16, 5120, 1280, 1), ("geglu32_640", 32, 32, 640, 5120, 1)]
    for name, H, W, cin, cout, ks in shapes:
        if args.only and not any(o in name for o in args.only.split(",")):
            continue
        trace_one(ops, lib, name, B, H, W, cin, cout, ks, args.split, flush_caches=not args.warm)


if __name__ == "__main__":
    main()
