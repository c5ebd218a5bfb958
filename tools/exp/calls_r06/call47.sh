#!/bin/bash
# round 6, call 47: two more seeds of the pair fuzz on the final code
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call47
mkdir -p $O
cd $R
export TMPDIR=/tmp
for cfg in "int8 110 3" "uint8 50 2"; do
  set -- $cfg
  timeout 300 python tools/fuzz_split.py --dtype $1 --seconds $2 --seed $3 2>&1 | grep -v "^Tengine" | tail -4 | cut -c1-700
done | tee $O/fuzz_split_device_more_seeds.txt
