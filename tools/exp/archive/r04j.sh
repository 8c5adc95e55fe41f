set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04j; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_text_encoder_gpu.py -m gpu -q -x -k "attention or encoder" 2>&1 | tail -5 > $OUT/pytest_attn.txt; cat $OUT/pytest_attn.txt
timeout 300 python tools/attn_bench.py --shapes "16,8,4096,40;16,8,1024,80;16,8,256,160;2,8,4096,40" --forms o3,o3s > $OUT/attn_bench.txt 2>&1; cat $OUT/attn_bench.txt
timeout 200 python tools/op_profile.py --model wukong --batch 16 --top 12 > $OUT/op_wukong.txt 2>&1; head -14 $OUT/op_wukong.txt
