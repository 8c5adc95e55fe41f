#!/bin/bash
# What bounds the fused transformer tail / head kernels at the 64x64 level of SDv2 (UNet batch 2: 256 blocks of 32 rows)?
# Micro-benchmark (cold weights, hipGraph) + 4 PMC passes of the same launches issued eagerly (rocprofv3 --pmc cannot follow graphs).
set -u
export TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r03sttail; mkdir -p $OUT
FUSED_ONLY=1 python tools/stchain_bench.py 2 4096 5 2>&1 | grep -v amdgpu.ids | tee $OUT/bench.txt
python tools/sthead_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench.txt
KSETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU")
for set in "${KSETS[@]}"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-50)
  NOGRAPH=1 FUSED_ONLY=1 NCOPY=24 timeout 200 rocprofv3 --pmc $set -d gpurun_out/pmcst/$tag -o pmc -- python tools/stchain_bench.py 2 4096 5 > $OUT/tail_$tag.log 2>&1 || echo "pass $tag failed"
done
echo "## st_tail_kernel<320, 1, 64> (32-row blocks): averages per dispatch" | tee $OUT/pmc.txt
python tools/pmc_kernel.py "st_tail_kernel<320, 1" $(find gpurun_out/pmcst -name "*_results.db") | tee -a $OUT/pmc.txt
echo "## st_tail_kernel<320, 2, 64> (64-row blocks): averages per dispatch" | tee -a $OUT/pmc.txt
python tools/pmc_kernel.py "st_tail_kernel<320, 2" $(find gpurun_out/pmcst -name "*_results.db") | tee -a $OUT/pmc.txt
rm -rf gpurun_out/pmcst
