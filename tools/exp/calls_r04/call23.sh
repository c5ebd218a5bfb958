#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call23
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python tools/exp/u8_dw_forms.py 1 2 4 0 2>&1 | grep -v "^Tengine" | tee $O/u8_dw_forms.txt
timeout 1200 python -m pytest tests/test_gpu_parity_uint8.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
timeout 600 python tools/exp/ab_step.py mssd 16 uint8 30 3 "one_row_blocks=TAMD_U8_DW_TH=1" "block_heights" 2>&1 | grep -v "^Tengine" | tee $O/ab_u8dw_mssd_b16.txt
