#!/bin/bash
# round 6, call 23: dwpw tests incl. the new 1024 / 2048-channel cases (the constants copy's tail loop, the LDS limit) and the pair fuzzer
# (tools/fuzz_pairs.py: random dwpw / pwdw pairs, fusion forced, random pwdw tile configurations, each graph also as two launches)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call23
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dwpw.py -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -25 > $O/pytest_dwpw.txt; tail -8 $O/pytest_dwpw.txt
for seed in 1 2 3; do
  timeout 400 python tools/fuzz_pairs.py --seconds 150 --seed $seed 2>&1 | grep -v "^Tengine" | tail -6 >> $O/fuzz_pairs_device.txt
done
cat $O/fuzz_pairs_device.txt | cut -c1-400
