#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06g
timeout 300 python tools/op_profile.py --batch 8 --latent 96 --top 400 2>&1 | grep -v amdgpu.ids > gpurun_out/r06g/op_profile_sd2_b8_l96.txt; head -60 gpurun_out/r06g/op_profile_sd2_b8_l96.txt
timeout 300 python tools/op_profile.py --model wukong --batch 16 --latent 64 --top 400 2>&1 | grep -v amdgpu.ids > gpurun_out/r06g/op_profile_wk_b16.txt; head -40 gpurun_out/r06g/op_profile_wk_b16.txt
