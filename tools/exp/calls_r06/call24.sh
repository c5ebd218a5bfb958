#!/bin/bash
# round 6, call 24: the GPU suite (own order + seed 9) + smoke + the driver's bench invocation on the final commit
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -40 > $O/pytest_gpu_all.txt; tail -3 $O/pytest_gpu_all.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Tengine" | tail -2 | tee -a $O/pytest_gpu_all.txt
timeout 1500 python tools/gpu_suite_shuffled.py 9 2>&1 | grep -v "^Tengine" > $O/s9.full.txt; (head -1 $O/s9.full.txt; tail -30 $O/s9.full.txt) > $O/pytest_gpu_shuffled_seed9.txt; rm $O/s9.full.txt; tail -2 $O/pytest_gpu_shuffled_seed9.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final_driver_invocation.json 2> $O/bench_final.err ) 2> $O/bench_final_time.txt; tail -c 400 $O/bench_final_driver_invocation.json; cat $O/bench_final_time.txt
