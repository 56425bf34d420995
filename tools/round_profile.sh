#!/bin/bash
# usage (on the GPU box): bash tools/round_profile.sh <tag> [notests]  — the round's measurements in one call:
# GPU tests, bench (f32 with the f64 / free-space / CPU legs), rocprofv3 kernel stats (four chunk streams and one),
# the two PMC traffic passes, the other BASELINE configs.  Results under gpurun_out/<tag>/.
# Every GPU step has its own time limit and the chain stops at the first failure.
tag=${1:-prof}
cd $GRAFT_REPO_ROOT && o=gpurun_out/$tag && mkdir -p $o && export TMPDIR=/tmp &&
if [ "$2" != "notests" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --durations=25 --timeout 300 --timeout-method thread > $o/pytest_gpu.log 2>&1; rc=$?
  tail -3 $o/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
fi &&
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err && echo bench ok &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof4 -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras > $o/bench_under_rocprof.json 2> $o/prof4.err && echo prof4 ok &&
GEAR_HIP_PIPE_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof1 -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras > $o/bench_under_rocprof_one_stream.json 2> $o/prof1.err && echo prof1 ok &&
GEAR_HIP_PIPE_STREAMS=1 bash tools/pmc_pass.sh 0x2003 $tag/pmc_fetch FETCH_SIZE &&
GEAR_HIP_PIPE_STREAMS=1 bash tools/pmc_pass.sh 0x2003 $tag/pmc_write WRITE_SIZE &&
timeout -k 10 300 python tools/bench_configs.py > $o/other_configs.txt 2> $o/other_configs.err && cat $o/other_configs.txt
