set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04b; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" 2>&1 | tail -5 > $OUT/pytest_attn.txt; cat $OUT/pytest_attn.txt
timeout 300 python tools/attn_bench.py --shapes "8,5,9216,64;2,5,4096,64;2,10,1024,64;2,20,256,64;16,8,4096,40;16,10,1024,64;8,10,2304,64" --forms o2,o3,o2s,o3s,o3s4 > $OUT/attn_bench.txt 2>&1; cat $OUT/attn_bench.txt
timeout 300 python tools/eval_ab.py --model sd2 --batch 2 --latent 64 --arms "r4a:attn_occ3=0,attn_kv_split=0" "occ3:attn_occ3=1,attn_kv_split=0" "occ3+split:attn_occ3=1,attn_kv_split=1" "occ2+split:attn_occ3=0,attn_kv_split=1" > $OUT/eval_ab_sd2_b2.txt 2>&1; grep -v amdgpu.ids $OUT/eval_ab_sd2_b2.txt
timeout 300 python tools/eval_ab.py --model sd2 --batch 8 --latent 96 --rounds 3 --iters 5 --arms "r4a:attn_occ3=0,attn_kv_split=0" "occ3:attn_occ3=1,attn_kv_split=0" > $OUT/eval_ab_sd2_768.txt 2>&1; grep -v amdgpu.ids $OUT/eval_ab_sd2_768.txt
timeout 300 python tools/eval_ab.py --model wukong --batch 16 --latent 64 --rounds 3 --iters 5 --arms "r4a:attn_occ3=0,attn_kv_split=0" "occ3:attn_occ3=1,attn_kv_split=0" > $OUT/eval_ab_wukong.txt 2>&1; grep -v amdgpu.ids $OUT/eval_ab_wukong.txt
