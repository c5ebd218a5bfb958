#!/bin/bash
# round 6, call 36: tests/test_gpu_split_batch.py after its fixes (full output), then the evidence script of call 35
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call36
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split_batch.py -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" > $O/pytest_split.txt; tail -60 $O/pytest_split.txt | cut -c1-300
bash tools/exp/calls_r06/call35.sh
