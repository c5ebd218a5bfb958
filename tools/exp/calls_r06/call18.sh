#!/bin/bash
# round 6, call 18: dwpw with its depthwise layer on the matrix cores (dwpw_body_mm): tests of both bodies, stage anatomy of both, and
# the committed build (vector-ALU depthwise, constants in LDS) against the tree's on MobileNet-v1 int8 b64
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call18
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dwpw.py tests/test_gpu_baseline_batches.py -m gpu -x -q --tb=short 2>&1 | grep -v "^Tengine" | tail -25 > $O/pytest_dwpw.txt; tail -12 $O/pytest_dwpw.txt
(echo "== depthwise on the matrix cores"; timeout 200 tools/exp/dwpw_anatomy.bin | head -8; echo "== depthwise on the vector ALU (TAMD_PIN=dwpw_mm=0)"; TAMD_PIN=dwpw_mm=0 timeout 200 tools/exp/dwpw_anatomy.bin | head -8) 2>&1 | tee $O/dwpw_anatomy.txt
AB_LAYERS=1 timeout 1500 python tools/exp/ab_lib.py mobilenet_v1 64 int8 200 3 valu=tools/exp/ab/ldsconsts.so mfma=product 2>&1 | grep -v "^Tengine" | tee $O/ab_mobilenet_b64_mm.txt | head -24
