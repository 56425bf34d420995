#!/bin/bash
# usage: tools/ab.sh [bench args...]   A/B of library builds / env settings inside ONE GPU call (box-to-box
# variance is ~10%): each line of tools/ab.cases is "label ENV=... ENV=..." run through bench.py
cd $GRAFT_REPO_ROOT
while read -r label envs; do
  [ -z "$label" ] && continue
  out=$(env $envs timeout -k 10 120 python bench.py --no-cpu-baseline --no-parity "$@" 2>&1 | tail -1)
  echo "$label $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f G/s  %.3f ms/step" % (d["value"]/1e9, d["roofline"]["kernel_ms"]))' 2>&1 | tail -1)"
done < tools/ab.cases
