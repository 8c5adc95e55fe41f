#!/usr/bin/env python
"""Matrix-pipe occupancy per kernel family from ONE rocprofv3 PMC pass of the bench command:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc/MFMA -o pmc -- \\
        python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph --no-other-configs
    python tools/pmc_mfma.py gpurun_out/pmc/MFMA/pmc_results.db "<command>" [attention launches per evaluation] > profiles/rNN_pmc_mfma.json

mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (sum SQ_BUSY_CYCLES x 32): SQ_BUSY_CYCLES is summed over the 32 shader engines, each
with 32 SIMDs (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per v_mfma_f32_32x32x16_f16) -- the share of
SIMD-cycles, while the shader engines were busy with the family's kernels, in which the matrix pipe was occupied.
mfma_busy_gui uses GRBM_GUI_ACTIVE / 8 x 1024 SIMDs as the denominator instead (whole-chip active time, which for 5-30 us
kernels includes the launch ramp: lower)."""
import json
import sqlite3
import sys
from collections import defaultdict

from pmc_traffic import family


def main():
    db = sqlite3.connect(sys.argv[1])
    command = sys.argv[2] if len(sys.argv) > 2 else "?"
    per_eval = float(sys.argv[3]) if len(sys.argv) > 3 else 32.0
    agg = defaultdict(lambda: defaultdict(float))
    rows = db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection")
    seen = defaultdict(set)
    for name, ctr, val, did in rows:
        fam = family(name)
        if not fam:
            continue
        agg[fam][ctr] += float(val)
        seen[fam].add(did)
    dur = defaultdict(float)
    try:
        for name, d in db.execute("select name, end - start from kernels"):
            fam = family(name)
            if fam:
                dur[fam] += float(d)
    except sqlite3.Error:
        pass
    evals = len(seen["attention"]) / per_eval if seen["attention"] else 1.0
    if seen.get("eval_marker"):      # one launch per UNet evaluation (tools/pmc_traffic.py family())
        evals = float(len(seen["eval_marker"]))
    agg.pop("eval_marker", None)
    seen.pop("eval_marker", None)
    dur.pop("eval_marker", None)
    out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- " + command,
           "unet_evals_in_trace": evals, "simds": 1024,
           "formula": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES * 32 SIMDs per shader engine); "
                      "mfma_busy_gui = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 1024)", "families": {}}
    for fam in sorted(agg):
        cyc = agg[fam].get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = agg[fam].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        sqb = agg[fam].get("SQ_BUSY_CYCLES", 0.0)
        ent = {"launches_per_eval": round(len(seen[fam]) / evals, 2),
               "mfma_busy": round(busy / (sqb * 32.0), 4) if sqb else None,
               "mfma_busy_gui": round(busy / (cyc * 1024.0), 4) if cyc else None,
               "mfma_busy_cycles_per_eval": round(busy / evals), "kernel_cycles_per_eval": round(cyc / evals)}
        if dur[fam] and cyc:
            ent["gui_cycles_per_ns"] = round(cyc / dur[fam], 3)
            ent["ms_per_eval_profiled"] = round(dur[fam] / evals / 1e6, 4)
        out["families"][fam] = ent
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
