#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
$R/tools/exp/launch_chain2.bin > $O/launch_chain2.txt 2>&1
cat $O/launch_chain2.txt
(echo "nproc $(nproc)"; echo "affinity $(taskset -p $$)"; echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; env | grep -i "omp\|thread" ; rocm-smi --showclocks 2>/dev/null | head -20) > $O/host_env.txt 2>&1
cat $O/host_env.txt
timeout 600 python -m pytest tests/test_gpu_baseline_batches.py -x -q -k depthwise > $O/pytest_dw_variants.txt 2>&1
tail -3 $O/pytest_dw_variants.txt
