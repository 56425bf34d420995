#!/bin/bash
# usage (GPU box): bash tools/stall_prof.sh <tag> [what]   — two SQ counter passes (issue/wait split, memory latencies)
tag=${1:-stall}; what=${2:-0x2003}
export GEAR_HIP_PIPE_STREAMS=1
bash $GRAFT_REPO_ROOT/tools/pmc_pass.sh $what $tag/a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS &&
bash $GRAFT_REPO_ROOT/tools/pmc_pass.sh $what $tag/b SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
bash $GRAFT_REPO_ROOT/tools/pmc_pass.sh $what $tag/c SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES &&
cd $GRAFT_REPO_ROOT && python tools/pmc_show.py $tag/a $tag/b $tag/c
