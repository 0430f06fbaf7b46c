#!/bin/bash
# PMC passes for one micro-benchmark (each --pmc set in its own run; kernel-trace only, as the pool requires)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
W=${1:-down3}
mkdir -p gpurun_out/pmc
rocprofv3 -L > gpurun_out/pmc/counters.txt 2>&1 || true
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc -o ${W}_p$i -- python3 tools/micro.py $W 5 > gpurun_out/pmc/${W}_p$i.log 2>&1 || echo "pass $i failed"
done
ls gpurun_out/pmc | head -40
