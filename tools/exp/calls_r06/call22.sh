#!/bin/bash
# round 6, call 22: device fuzz on the final code (after the dwpw rework and the read-once fix): int8, uint8, heads -- three seeds each
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call22
mkdir -p $O
cd $R
export TMPDIR=/tmp
for seed in 31 32 33; do
  timeout 260 python tools/fuzz_heads.py --seconds 100 --seed $seed 2>&1 | grep -v "^Tengine" | tail -2 >> $O/fuzz_heads_device.txt
  timeout 260 python tools/fuzz_device.py --dtype int8 --seconds 150 --seed $((seed + 100)) 2>&1 | grep -v "^Tengine" | tail -3 >> $O/fuzz_device_int8.txt
  timeout 260 python tools/fuzz_device.py --dtype uint8 --seconds 100 --seed $((seed + 200)) 2>&1 | grep -v "^Tengine" | grep -v "^kernels exercised" | tail -2 >> $O/fuzz_device_uint8.txt
done
cat $O/fuzz_heads_device.txt $O/fuzz_device_int8.txt $O/fuzz_device_uint8.txt
