#!/bin/bash
# PMC profile of the shaded renderer in adjoint mode (and finite differences for comparison), 32x4 + grid
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for W in c32l4_grid16_relu c64l6_grid16_relu; do
for M in 2 1; do
  O=$R/gpurun_out/prof_shaded_${W}_mode$M; mkdir -p $O
  i=0
  for set in \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES" \
    "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR" \
    "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT" \
    "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCC_HIT TCC_MISS"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/tools/dev/shaded_one.py $W $M 2 > $O/pmc$i.log 2>&1
  done
  python3 $R/tools/pmc_summary.py $O shaded_${W}_mode$M > $R/gpurun_out/shaded_${W}_mode${M}_pmc.csv
  grep "ms / frame" $O/pmc1.log
done
done
