#!/bin/bash
# what bounds the self-attention kernel at the SD 768 shape (UNet batch 8, 5 heads, 9216 tokens, d = 64)?
set -u
export TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02attn; mkdir -p $OUT
python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_bench.txt
KSETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU")
for set in "${KSETS[@]}"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-50)
  timeout 120 rocprofv3 --pmc $set -d gpurun_out/pmca/$tag -o pmc -- python tools/attn_bench.py --shapes "8,5,9216,64" --iters 3 > $OUT/attn_$tag.log 2>&1 || echo "pass $tag failed"
done
python tools/pmc_kernel.py attn_kernel $(find gpurun_out/pmca -name "*_results.db") | tee $OUT/attn_pmc.txt
rm -rf gpurun_out/pmca
