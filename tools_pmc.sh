#!/bin/bash
# usage: tools_pmc.sh <what> <outdir-name> <counters...>
what=$1; out=$2; shift 2
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/tools_ablate1.py $what > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/$out/err.log
echo rc=$?
