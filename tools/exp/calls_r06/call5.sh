#!/bin/bash
# round 6, call 5: the new tests (single-input concat, fp32 single-hop against the real reference), device fuzz: the head plumbing and
# the general int8 / uint8 campaigns on the re-split translation units
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call5
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_int8_heads.py tests/test_gpu_parity_fp32.py tests/test_gpu_parity_uint8.py -m gpu -q --tb=short -s 2>&1 | grep -v "^Tengine" > $O/tests.txt; grep "fp32 parity device" $O/tests.txt | cut -c1-200; tail -8 $O/tests.txt | cut -c1-300
timeout 200 python tools/fuzz_heads.py --seconds 90 --seed 11 2>&1 | grep -v "^Tengine" | tee $O/fuzz_heads_device.txt | tail -3
timeout 200 python tools/fuzz_device.py --dtype int8 --seconds 90 --seed 61 2>&1 | grep -v "^Tengine" | tee $O/fuzz_device_int8.txt | tail -3
timeout 200 python tools/fuzz_device.py --dtype uint8 --seconds 90 --seed 62 2>&1 | grep -v "^Tengine" | tee $O/fuzz_device_uint8.txt | tail -3
