#!/bin/bash
# round 5, call 30: one graph = one thread at a time, enforced: the async / direct / plugin / distributed tests, then batch 1 against the evidence build
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call30
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_async.py tests/test_gpu_direct.py tests/test_plugin_dropin.py tests/test_tm_benchmark_hip.py tests/test_gpu_bench_dist.py tests/test_gpu_rccl_c.py tests/test_reference_benchmark_files.py -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -25
timeout 600 python tools/exp/ab_lib.py mobilenet_v1 1 int8 2000 3 evidence=$R/tools/exp/ab/libtengine_amd_r05_final_evidence.so guarded=product 2>&1 | grep -v "^Tengine" | tee $O/ab_one_thread_guard_mobilenet_v1_b1.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
