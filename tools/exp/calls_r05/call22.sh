#!/bin/bash
# round 5, call 22: the whole GPU suite and the device fuzz on the final kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call22
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu_all.txt; tail -3 $O/pytest_gpu_all.txt
timeout 400 python tools/fuzz_device.py --dtype int8 --seconds 100 --seed 7 > $O/fuzz_device_int8.txt 2>&1; tail -2 $O/fuzz_device_int8.txt
timeout 400 python tools/fuzz_device.py --dtype uint8 --seconds 80 --seed 7 > $O/fuzz_device_uint8.txt 2>&1; tail -2 $O/fuzz_device_uint8.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Tengine" | tail -3
