#!/bin/bash
# round 6, call 7: the pwdw / head tests after the by-construction fuse went back to batch 1 only; the evidence pass again (call 6's plans
# had fused MobileNet-v1 b64's 7x7 tail by construction: 287 instead of 265 us)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pwdw.py tests/test_gpu_dwpw.py tests/test_gpu_int8_heads.py tests/test_gpu_baseline_batches.py -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -8
bash tools/collect_evidence_r06.sh r06 2>&1 | tee $O/collect.log | tail -150
