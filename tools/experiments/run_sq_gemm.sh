cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/psq
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/psq -o x -- python $R/tools/bench_panel.py 65536 > /dev/null 2> /tmp/psq.err || tail -3 /tmp/psq.err
  python $R/tools/rocpd_sq.py $(find /tmp/psq -name '*.db' | head -1) "gemm256_kernel<0, false, false>"
done
