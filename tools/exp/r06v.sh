#!/bin/bash
# round 6, pass V: eight-wave form of the 128-row HALO patch tile -- parity, then the in-situ tuner with the new candidates (sd2 batch 2 and the others)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06v
mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "eight_wave_patch or conv3x3" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
bash tools/retune_insitu.sh 2>&1 | tail -70
