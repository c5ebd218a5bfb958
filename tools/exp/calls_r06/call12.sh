#!/bin/bash
# round 6, call 12: longer device fuzz campaigns on the final code (spare GPU budget): heads, int8, uint8 -- four seeds each
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call12
mkdir -p $O
cd $R
export TMPDIR=/tmp
for seed in 21 22 23 24; do
  timeout 260 python tools/fuzz_heads.py --seconds 120 --seed $seed 2>&1 | grep -v "^Tengine" | tail -2 >> $O/fuzz_heads_device.txt
  timeout 260 python tools/fuzz_device.py --dtype int8 --seconds 120 --seed $((seed + 100)) 2>&1 | grep -v "^Tengine" | grep -v "^kernels exercised" | tail -2 >> $O/fuzz_device_int8.txt
  timeout 260 python tools/fuzz_device.py --dtype uint8 --seconds 120 --seed $((seed + 200)) 2>&1 | grep -v "^Tengine" | grep -v "^kernels exercised" | tail -2 >> $O/fuzz_device_uint8.txt
done
cat $O/fuzz_heads_device.txt $O/fuzz_device_int8.txt $O/fuzz_device_uint8.txt
