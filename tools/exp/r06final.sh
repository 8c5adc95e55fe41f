#!/bin/bash
# round 6, final-tree pass: suite + smoke, bench (all configs), kernel traces, op profiles, matrix-pipe PMC of the batch >= 8 configs
set -u
export TMPDIR=/tmp
bash tools/gpu_pass.sh r06i tests bench prof opprof profcfg pmc_mfma_cfg
timeout 200 python tools/op_profile.py --model wukong --batch 16 --latent 64 --guidance --top 60 > gpurun_out/r06i/op_profile_wukong_b16.txt 2>&1; head -6 gpurun_out/r06i/op_profile_wukong_b16.txt
timeout 200 python tools/op_profile.py --model sd2 --batch 8 --latent 96 --guidance --top 60 > gpurun_out/r06i/op_profile_sd2_b8_l96.txt 2>&1; head -6 gpurun_out/r06i/op_profile_sd2_b8_l96.txt
