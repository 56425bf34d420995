#!/bin/bash
# usage (GPU box): bash tools/quick_prof.sh <tag> [bench args]   — one-stream kernel stats + one SQ counter pass
tag=${1:-qp}; shift
cd $GRAFT_REPO_ROOT && o=gpurun_out/$tag && mkdir -p $o && export TMPDIR=/tmp &&
GEAR_HIP_PIPE_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof1 -o p -- python bench.py --no-cpu-baseline --no-parity "$@" > $o/bench_one_stream.json 2> $o/prof1.err &&
python - <<PY
import csv,glob
f=glob.glob("$o/prof1/*kernel_stats.csv")+glob.glob("$o/prof1/*/*kernel_stats.csv")
for r in csv.DictReader(open(f[0])):
    if "gear_hip" in r["Name"]: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
GEAR_HIP_PIPE_STREAMS=1 bash tools/pmc_pass.sh 0x2003 $tag/sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU &&
python tools/pmc_show.py $tag/sq1
