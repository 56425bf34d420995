#!/bin/bash
# usage (GPU box): bash tools/attrib_sweep.sh <tag>
# Per-region instruction attribution of the sweep kernel by differences between ablated runs of the
# -DGEAR_HIP_ABLATE build (build/libgear_hip_ablate.so): one rocprofv3 --pmc pass of SQ counters per variant
# over tools/run_once.py (512k configurations of the metric scene, one stream).  tools/attrib_show.py
# turns the passes into the table committed under profiles/.
# Two more passes over the unablated run give the issue-side picture of the shipped kernel: busy cycles of the
# sequencers and GRBM_GUI_ACTIVE (the clock the chip held), and the memory-instruction counts.
tag=${1:-attrib}
export GEAR_HIP_PIPE_STREAMS=1
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
run() {  # label what lib [counters]
  export GEAR_HIP_LIB=$GRAFT_REPO_ROOT/build/$3
  bash $GRAFT_REPO_ROOT/tools/pmc_pass.sh $2 $tag/$1 ${4:-$C} || exit 1
}
run full        0x2003   libgear_hip_ablate.so &&
run busy        0x2003   libgear_hip_ablate.so "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" &&
run mem         0x2003   libgear_hip_ablate.so "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE" &&
run no_stage1   0x2203   libgear_hip_ablate.so &&
run no_obb      0x202003 libgear_hip_ablate.so &&
run no_push     0x2403   libgear_hip_ablate.so &&
run no_push_obb 0x202403 libgear_hip_ablate.so &&
run fk_self     0x2c03   libgear_hip_ablate.so &&
run fk_only     0x2c01   libgear_hip_ablate.so &&
run env_only    0x2001   libgear_hip_ablate.so &&
run self_only   0x2002   libgear_hip_ablate.so
