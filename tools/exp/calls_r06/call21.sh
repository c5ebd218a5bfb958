#!/bin/bash
# round 6, call 21: after the TAMD_AUTOTUNE / TAMD_FUSE_ELTWISE read-once fix (call 20's seed-7 order failed three plan-cache tests): the
# suite in the failing order, one more seed, its own order, smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
export TMPDIR=/tmp
for seed in 7 8; do
  timeout 1500 python tools/gpu_suite_shuffled.py $seed 2>&1 | grep -v "^Tengine" > $O/pytest_gpu_shuffled_seed$seed.full.txt
  (head -1 $O/pytest_gpu_shuffled_seed$seed.full.txt; tail -30 $O/pytest_gpu_shuffled_seed$seed.full.txt) > $O/pytest_gpu_shuffled_seed$seed.txt; rm $O/pytest_gpu_shuffled_seed$seed.full.txt
  head -1 $O/pytest_gpu_shuffled_seed$seed.txt | cut -c1-200; tail -2 $O/pytest_gpu_shuffled_seed$seed.txt
done
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -40 > $O/pytest_gpu_all.txt; tail -3 $O/pytest_gpu_all.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Tengine" | tail -2 | tee -a $O/pytest_gpu_all.txt
