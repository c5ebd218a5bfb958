#!/bin/bash
# round 6, call 25: a longer pair-fuzz campaign on the final commit (six more seeds)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call25
mkdir -p $O
cd $R
export TMPDIR=/tmp
for seed in 4 5 6 7 8 9; do
  timeout 400 python tools/fuzz_pairs.py --seconds 200 --seed $seed 2>&1 | grep -v "^Tengine" | tail -6 >> $O/fuzz_pairs_device.txt
done
cut -c1-200 $O/fuzz_pairs_device.txt
