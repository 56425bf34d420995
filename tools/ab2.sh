#!/bin/bash
# usage: tools/ab2.sh <cases-file>   each line: "label ENV=.. ENV=.. -- args of tools/tp.py"; all inside ONE GPU call
cd $GRAFT_REPO_ROOT
while read -r line; do
  [ -z "$line" ] && continue
  label=${line%% *}; rest=${line#* }
  envs=${rest%%--*}; args=${rest#*--}
  out=$(env $envs timeout -k 10 120 python tools/tp.py $args 2>&1 | tail -1)
  echo "$label | $out"
done < $1
