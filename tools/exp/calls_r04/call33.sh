#!/bin/bash
# round 4, call 33: device fuzz on the final code, every round-4 form among the randomly pinned members
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call33
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 200 python tools/fuzz_device.py --dtype uint8 --seconds 70 --seed 4 2>&1 | grep -v "^Tengine" | tail -4 | tee $O/fuzz_device_uint8.txt
timeout 200 python tools/fuzz_device.py --dtype int8 --seconds 50 --seed 4 2>&1 | grep -v "^Tengine" | tail -4 | tee $O/fuzz_device_int8.txt
