#!/bin/bash
# round 5, call 27: small pointwise + tail pairs fused by construction (no throughput race): six fresh batch-1 plans in a row, the pwdw / plan tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call27
mkdir -p $O
cd $R
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  TAMD_PLAN_CACHE=$O/plan_$i.txt python tools/run_model.py mobilenet_v1 1 200 int8 2>&1 | grep -v "^Tengine" | tail -1
done | tee $O/six_fresh_batch1_plans.txt
rm -f $O/plan_*.txt
timeout 900 python -m pytest tests/test_gpu_pwdw.py tests/test_gpu_plan_cache.py tests/test_gpu_baseline_batches.py tests/test_gpu_direct.py -m gpu -x -q 2>&1 | tail -3
