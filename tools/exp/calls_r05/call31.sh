#!/bin/bash
# round 5, call 31: the whole GPU suite on the final code (short tracebacks kept), smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call31
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -80 > $O/pytest_gpu_all.txt; tail -4 $O/pytest_gpu_all.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Tengine" | tail -2
