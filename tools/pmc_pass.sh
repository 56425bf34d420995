#!/bin/bash
# usage: tools/pmc_pass.sh <what> <outdir-name> <counters...>   (one rocprofv3 --pmc pass over tools/run_once.py)
what=$1; out=$2; shift 2
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/tools/run_once.py $what > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/$out/err.log
echo rc=$?
