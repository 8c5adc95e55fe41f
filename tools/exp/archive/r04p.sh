set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04p; mkdir -p $OUT
timeout 300 python tools/eval_ab.py --model wukong --batch 16 --latent 64 --rounds 3 --iters 5 --arms "p0:gemm_dense8p=0" "p2:gemm_dense8p=2" > $OUT/eval_ab_wukong.txt 2>&1; grep -v amdgpu.ids $OUT/eval_ab_wukong.txt
timeout 300 python tools/eval_ab.py --model sd2 --batch 8 --latent 96 --rounds 3 --iters 5 --arms "p0:gemm_dense8p=0" "p2:gemm_dense8p=2" > $OUT/eval_ab_sd2_768.txt 2>&1; grep -v amdgpu.ids $OUT/eval_ab_sd2_768.txt
timeout 300 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --arms "p0:gemm_dense8p=0" "p2:gemm_dense8p=2" > $OUT/eval_ab_sd2_b2.txt 2>&1; grep -v amdgpu.ids $OUT/eval_ab_sd2_b2.txt
