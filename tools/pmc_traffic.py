#!/usr/bin/env python
"""Aggregate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate --pmc runs of tools/op_profile.py) into HBM
bytes per kernel family per UNet evaluation.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc/FETCH_SIZE -o pmc -- python tools/op_profile.py --batch 2 --passes 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc/WRITE_SIZE -o pmc -- python tools/op_profile.py --batch 2 --passes 1
    python tools/pmc_traffic.py gpurun_out/pmc > profiles/rNN_pmc_traffic.json

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950
the FETCH_SIZE expression tallies 128-B read requests at 64 B, so it is DOUBLED before it is compared with byte counts.
Infinity-Cache (MALL) hits are included in both counters: this is traffic at the L2 <-> fabric boundary, an upper
bound on HBM traffic.
"""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def family(name):
    if name.startswith("void at::") or "at::native" in name or "rocclr" in name:
        return None   # torch's weight-initialisation / copy kernels
    if "nchw_to_nhwc_kernel" in name:
        return "eval_marker"      # exactly one launch per UNet evaluation (LatentDiffusion.apply_model's layout boundary)
    if ("gemm_kernel" in name or "dense_kernel" in name or "conv3x3_halo_kernel" in name or "st_tail_kernel" in name
            or "st_head_kernel" in name or "conv8p_kernel" in name):
        return "gemm"      # (the fused SpatialTransformer head / tail launches are chains of dense GEMMs)
    if "splitk_reduce" in name:
        return "splitk_reduce"
    if "attn_kernel" in name or "attn_pipe_kernel" in name or "attn8_kernel" in name:
        return "attention"
    if "gn_stats" in name or "gn_apply" in name or "gn_fused" in name:
        return "groupnorm"
    if "ln_kernel" in name:
        return "layernorm"
    if "GLOBAL__N" in name or "anonymous namespace" in name:
        return "other_mdx"
    return None   # torch initialisation kernels etc.


def load(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,))
    agg = defaultdict(lambda: [0, 0.0])
    for name, val in rows:
        fam = family(name)
        if fam:
            agg[fam][0] += 1
            agg[fam][1] += float(val)
    return agg


def main():
    root = sys.argv[1]
    command = sys.argv[2] if len(sys.argv) > 2 else "python tools/op_profile.py --batch 2 --passes 1"
    fetch = load(f"{root}/FETCH_SIZE/pmc_results.db", "FETCH_SIZE")
    write = load(f"{root}/WRITE_SIZE/pmc_results.db", "WRITE_SIZE")
    # attention launches per SDv2 UNet evaluation: 32, or 27 when the five 64 x 64 cross-attentions run inside the fused tails
    per_eval = float(sys.argv[3]) if len(sys.argv) > 3 else 32.0
    evals = fetch["attention"][0] / per_eval if fetch["attention"][0] else 1.0
    if fetch.get("eval_marker", [0])[0]:      # (round 6: the attention count per evaluation depends on which cross-attentions ride on their
        evals = float(fetch["eval_marker"][0])       # projections; the layout-boundary kernel runs exactly once per evaluation)
    for agg in (fetch, write):
        agg.pop("eval_marker", None)
    out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- " + command,
           "unet_evals_in_trace": evals,
           "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B); MALL hits included",
           "families": {}}
    tot_r = tot_w = 0.0
    for fam in sorted(set(fetch) | set(write)):
        n = fetch[fam][0] or write[fam][0]
        rd = fetch[fam][1] * 1024.0 * 2.0 / evals
        wr = write[fam][1] * 1024.0 / evals
        tot_r += rd
        tot_w += wr
        out["families"][fam] = {"launches_per_eval": n / evals, "read_MB_per_eval": round(rd / 1e6, 1),
                                "write_MB_per_eval": round(wr / 1e6, 1),
                                "bytes_per_launch": int((rd + wr) / max(1.0, n / evals))}
    out["total_read_MB_per_eval"] = round(tot_r / 1e6, 1)
    out["total_write_MB_per_eval"] = round(tot_w / 1e6, 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
