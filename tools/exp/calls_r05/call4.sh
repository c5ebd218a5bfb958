#!/bin/bash
# round 5, call 4: after the old unrolled-taps kernel left the library -- parity of the product's 3x3 forms and of the GEMM family,
# the reference's own benchmark files through the plugin, the plugin / baseline-batch tests the YOLOv3-tiny builder change touches;
# final anatomy (all forms, experiments included); PMC pass 2 (MFMA busy, LDS bank conflicts) of ResNet-50 b32 on a fresh plan
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call4
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_pgemm.py tests/test_reference_benchmark_files.py tests/test_gpu_gemm_family.py tests/test_gpu_direct.py tests/test_plugin_dropin.py tests/test_gpu_baseline_batches.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt | grep -v "^Tengine"
timeout 300 tools/exp/pgemm_anatomy.bin 32 3x3 > $O/pgemm_anatomy_3x3.txt 2>&1
grep -c "us/launch" $O/pgemm_anatomy_3x3.txt
export TAMD_PLAN_CACHE=$O/plan_resnet50_int8_b32.txt
timeout 300 python tools/profile_layers.py resnet50 32 20 int8 2>&1 | grep -v "^Tengine" > $O/layers_resnet50_int8_b32.txt
tail -1 $O/layers_resnet50_int8_b32.txt
cd /tmp
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2 -- python $R/tools/run_model.py resnet50 32 2 int8 > $O/pmc2.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_mfma_resnet50_int8_b32.csv $O/pmc2 > /dev/null
rm -rf $O/pmc2
grep -i "pgemm" $O/pmc_mfma_resnet50_int8_b32.csv | cut -c1-200
