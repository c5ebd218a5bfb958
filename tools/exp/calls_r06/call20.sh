#!/bin/bash
# round 6, call 20: the evidence pass (plans, PMC, bench lines, rocprofv3 stats, layer tables) + HSA dispatch stamps + the GPU suite in two
# orders + smoke on the code after the dwpw rework
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
export TMPDIR=/tmp
bash tools/collect_evidence_r06.sh r06 2>&1 | tee $O/collect.log | grep -v "^/" | head -60
cd $R
for cfg in "mobilenet_v1 1 int8" "mobilenet_v1 64 int8" "resnet50 32 int8" "yolov3_tiny 8 uint8" "mssd 16 uint8"; do
  set -- $cfg
  timeout 300 python tools/direct_timestamps.py $1 $2 $3 500 2>&1 | grep -v "^Tengine" > $O/direct_path_timestamps_$1_$3_b$2.txt; tail -2 $O/direct_path_timestamps_$1_$3_b$2.txt
done
timeout 1500 python tools/gpu_suite_shuffled.py 7 2>&1 | grep -v "^Tengine" | tail -40 > $O/pytest_gpu_shuffled_seed7.txt; head -1 $O/pytest_gpu_shuffled_seed7.txt | cut -c1-150; tail -2 $O/pytest_gpu_shuffled_seed7.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -40 > $O/pytest_gpu_all.txt; tail -3 $O/pytest_gpu_all.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Tengine" | tail -2 | tee -a $O/pytest_gpu_all.txt
