#!/bin/bash
# round 6, call 11: the evidence pass and the GPU suite (own order + one seeded random order) + smoke on the FINAL code
# (after tamd_graph_direct_timestamps / roofline.hsa_dispatch_stamps)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
export TMPDIR=/tmp
bash tools/collect_evidence_r06.sh r06 2>&1 | tee $O/collect.log | grep -v "^/" | head -60
cd $R
timeout 1500 python tools/gpu_suite_shuffled.py 6 2>&1 | grep -v "^Tengine" | tail -40 > $O/pytest_gpu_shuffled_seed6.txt; head -1 $O/pytest_gpu_shuffled_seed6.txt | cut -c1-150; tail -2 $O/pytest_gpu_shuffled_seed6.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -40 > $O/pytest_gpu_all.txt; tail -3 $O/pytest_gpu_all.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Tengine" | tail -2 | tee -a $O/pytest_gpu_all.txt
