#!/bin/bash
# usage (on the GPU box): bash tools/small_profile.sh <tag> — the small-batch measurements of profiles/<round>/small_batch.txt:
# k_small / pipeline / fused latencies, other scenes, rocprofv3 kernel durations of k_small, plan() wall time
tag=${1:-small}
cd $GRAFT_REPO_ROOT && o=gpurun_out/$tag && mkdir -p $o && export TMPDIR=/tmp &&
( timeout -k 10 100 python tools/small_batch.py 0x0 1,64,256,1024,2048,4096,8192,16000 32 &&
  timeout -k 10 100 python tools/small_batch.py 0x0 64,1024,4096 64 &&
  timeout -k 10 100 python tools/small_batch.py 0x2000 1,64,1024,4096,8192,16000 32 &&
  timeout -k 10 100 python tools/small_batch.py 0x1000 64,1024 32 &&
  timeout -k 10 100 python tools/small_batch.py 0xffff 64 32 &&
  for wl in pr2 sawyer ur10; do
    timeout -k 10 100 python tools/small_batch.py 0x20 64,1024,4096,8192,16384,32768 32 $wl &&
    timeout -k 10 100 python tools/small_batch.py 0x2000 64,1024,4096,8192,16384,32768 32 $wl || exit 1
  done ) > $o/latency.txt 2>&1 && echo latency ok &&
for B in 1 64 256 1024 4096; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$B -o p -- python tools/small_batch.py 0x0 $B 32 > /dev/null 2> $o/prof_$B.err || exit 1
  grep k_small $o/prof_$B/p_kernel_stats.csv | head -1 | sed "s/^/B=$B /" >> $o/kernel_durations.txt
done && echo prof ok &&
timeout -k 10 200 python tools/plan_time.py > $o/plan_time.txt 2>&1 && cat $o/plan_time.txt
