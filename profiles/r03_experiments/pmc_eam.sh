#!/bin/bash
# pmc_eam.sh TAG "ENVS": the usual counter groups for EAM cta_cell at 80^3 (separate --pmc passes, --kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/profiles/pmc_run.sh "$1" "$2" "--pot eam --method cta_cell --steps 6 --warmup 2" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
  "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"
