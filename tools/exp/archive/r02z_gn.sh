#!/bin/bash
# blocks per column-statistics GroupNorm launch (each block re-folds its channels' row-block partials).  Record of a finished experiment:
# the MDX_GN_CS_BLOCKS knob it drove was removed again (no setting beat 1024)
export PYTHONPATH=.
mkdir -p gpurun_out/r02z
for t in 1024 256 512 2048 1024; do
  MDX_GN_CS_BLOCKS=$t python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gn blocks $t', r['value'], r['per_unet_step_ms'], r['roofline']['families']['groupnorm'])"
done | tee gpurun_out/r02z/gn.txt
