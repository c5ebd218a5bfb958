#!/bin/bash
# round 6, call 1: reproduce round 5's one unexplained failure (test_device_copy_of_the_outputs_after_a_zero_copy_run after
# test_gpu_plan_cache.py): call 27's exact subset with full tracebacks (up to 3 times), then the labelled loop of tools/exp/zc_flake.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call1
mkdir -p $O
cd $R
export TMPDIR=/tmp
for i in 1 2; do
  timeout 900 python -m pytest tests/test_gpu_pwdw.py tests/test_gpu_plan_cache.py tests/test_gpu_baseline_batches.py tests/test_gpu_direct.py -m gpu -x -q --tb=long 2>&1 | grep -v "^Tengine" > $O/subset_$i.txt
  tail -2 $O/subset_$i.txt
  grep -q "failed" $O/subset_$i.txt && break
done
for i in 1 2; do
  timeout 600 python -m pytest tests/test_gpu_plan_cache.py tests/test_gpu_direct.py -m gpu -x -q --tb=long 2>&1 | grep -v "^Tengine" > $O/pair_$i.txt
  tail -2 $O/pair_$i.txt
done
timeout 900 python tools/exp/zc_flake.py 40 1 2>&1 | grep -v "^Tengine" | tee $O/zc_flake_pc1.txt | tail -15
timeout 600 python tools/exp/zc_flake.py 40 0 2>&1 | grep -v "^Tengine" | tee $O/zc_flake_pc0.txt | tail -15
python bench.py --steps 20 --warmup 5 > $O/bench_b1_driver_invocation.json 2> $O/bench_b1.err; tail -c 600 $O/bench_b1_driver_invocation.json
