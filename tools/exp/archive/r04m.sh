set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04m; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_glide_gpu.py -m gpu -q -x -k "groupnorm or gn or film or resblock" 2>&1 | tail -4 > $OUT/pytest_gn.txt; cat $OUT/pytest_gn.txt
timeout 300 python tools/eval_ab.py --model wukong --batch 16 --latent 64 --rounds 3 --iters 5 --arms "f0:gn_fused_small=0" "f1:gn_fused_small=1" > $OUT/eval_ab_wukong.txt 2>&1; grep -v amdgpu.ids $OUT/eval_ab_wukong.txt
timeout 300 python tools/eval_ab.py --model sd2 --batch 8 --latent 96 --rounds 3 --iters 5 --arms "f0:gn_fused_small=0" "f1:gn_fused_small=1" > $OUT/eval_ab_sd2_768.txt 2>&1; grep -v amdgpu.ids $OUT/eval_ab_sd2_768.txt
for a in 0 1 1 0; do MDX_GN_FUSED_SMALL=$a timeout 300 python bench.py --config glide_256 --no-cpu-baseline --steps 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('glide gn_fused_small=$a', d['value'], d['unit'])"; done
