#!/bin/bash
# round 5, call 16: four slices per pwdw block: parity, then MobileNet-v1 b64 / b1 against the build before the round's requantisation work
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call16
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_pwdw.py tests/test_gpu_baseline_batches.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_subset.txt
PRE=$R/tools/exp/ab/libtengine_amd_r05_pre_window.so
for cfg in "mobilenet_v1 64 int8 100 2" "mobilenet_v1 1 int8 2000 3"; do
  set -- $cfg
  AB_LAYERS=1 timeout 900 python tools/exp/ab_lib.py $1 $2 $3 $4 $5 before=$PRE now=product > $O/ab_slices4_layers_$1_b$2.txt 2>&1
  grep -v "^Tengine" $O/ab_slices4_layers_$1_b$2.txt | cut -c1-160
done
