#!/bin/bash
# round 6, call 42: random even-batch graphs as two half-batch device graphs behind one handle, every run path, against the oracle
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call42
mkdir -p $O
cd $R
export TMPDIR=/tmp
for cfg in "int8 80 1" "int8 60 2" "uint8 60 1"; do
  set -- $cfg
  timeout 400 python tools/fuzz_split.py --dtype $1 --seconds $2 --seed $3 2>&1 | grep -v "^Tengine" | tail -6
done | tee $O/fuzz_split_device.txt
