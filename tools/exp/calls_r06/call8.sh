#!/bin/bash
# round 6, call 8: the GPU suite on the final code -- three seeded random file orders and the suite's own order -- and smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call8
mkdir -p $O
cd $R
export TMPDIR=/tmp
for seed in 3 4 5; do
  timeout 1500 python tools/gpu_suite_shuffled.py $seed 2>&1 | grep -v "^Tengine" | tail -40 > $O/pytest_gpu_shuffled_seed$seed.txt; head -1 $O/pytest_gpu_shuffled_seed$seed.txt | cut -c1-200; tail -2 $O/pytest_gpu_shuffled_seed$seed.txt
done
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -40 > $O/pytest_gpu_all.txt; tail -3 $O/pytest_gpu_all.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Tengine" | tail -2 | tee $O/smoke.txt
