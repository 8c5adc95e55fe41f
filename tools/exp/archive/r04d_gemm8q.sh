set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gemm8p_gpu.py -m gpu -q -x 2>&1 | tail -5 > $OUT/pytest_gemm8p.txt; cat $OUT/pytest_gemm8p.txt
MDX_GEMM_DENSE8Q_VAR=32 timeout 900 python -m pytest tests/test_gemm8p_gpu.py -m gpu -q -x 2>&1 | tail -5 > $OUT/pytest_gemm8p_rb.txt; cat $OUT/pytest_gemm8p_rb.txt


timeout 300 python tools/eval_ab.py --model wukong --batch 16 --latent 64 --rounds 3 --iters 5 --arms "q0:gemm_dense8q=0" "q1rb:gemm_dense8q=1,gemm_dense8q_var=32" > $OUT/eval_ab_wukong.txt 2>&1; grep -v amdgpu.ids $OUT/eval_ab_wukong.txt
timeout 300 python tools/eval_ab.py --model sd2 --batch 8 --latent 96 --rounds 3 --iters 5 --arms "q0:gemm_dense8q=0" "q1rb:gemm_dense8q=1,gemm_dense8q_var=32" > $OUT/eval_ab_sd2_768.txt 2>&1; grep -v amdgpu.ids $OUT/eval_ab_sd2_768.txt
timeout 300 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --arms "a:gemm_dense8q=0" "b:gemm_dense8q=0" > $OUT/eval_ab_sd2_b2.txt 2>&1; grep -v amdgpu.ids $OUT/eval_ab_sd2_b2.txt
