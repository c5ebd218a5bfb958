#!/bin/bash
# round 4, call 36: MobileNet-v1 b64 evidence refreshed after dwpw (plan, PMC traffic / MFMA busy, bench line, layer table) -- the other
# configurations are untouched by that change
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call36
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P=$R/tengine_amd/plans/mobilenet_v1_int8_b64.txt
rm -f $P
TAMD_PLAN_CACHE=$P python - <<PY 2>&1 | grep -v "^Tengine"
import sys; sys.path.insert(0, "$R")
from tengine_amd import capi, models, tm2
g = models.build("mobilenet_v1", "int8", 64)
gr = capi.Graph(tm2.write_tm2(g), batch=64, direct_dispatch=True)
gr.set_input(models.synth_input(g, 3, tm2.DT_INT8)); gr.run(); gr.upload(); gr.sync(); gr.time_launches(20)
print("mobilenet_v1 int8 b64: %d launches, %.1f us/step" % (gr.kernel_num(), 1e3 * gr.time_launches(100) / 100))
gr.close()
PY
cp $P $O/plan_mobilenet_v1_int8_b64.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/calib_$c -- $R/tools/exp/hbm_calib.bin > $O/calib_$c.log 2>&1
  cp $P $O/pmc_plan.txt
  TAMD_PLAN_CACHE=$O/pmc_plan.txt rocprofv3 --pmc $c --output-format csv -d $O/m_$c -- python $R/tools/run_model.py mobilenet_v1 64 5 int8 > $O/m_$c.log 2>&1
done
K=$(grep -o "launches_per_run [0-9]*" $O/m_FETCH_SIZE.log | cut -d' ' -f2)
python $R/tools/traffic_summary.py $O/traffic_mobilenet_v1_int8_b64.json $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE $O/m_FETCH_SIZE $O/m_WRITE_SIZE $((K * 5)) > $O/traffic_mobilenet_v1_int8_b64.txt 2>&1
cp $O/traffic_mobilenet_v1_int8_b64.json $R/profiles/r04_traffic_mobilenet_v1_int8_b64.json
TAMD_PLAN_CACHE=$O/pmc_plan.txt rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/m_mfma -- python $R/tools/run_model.py mobilenet_v1 64 5 int8 > $O/m_mfma.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_mfma_mobilenet_v1_int8_b64.csv $O/m_mfma > /dev/null 2>&1
python $R/bench.py --model mobilenet_v1 --batch 64 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_mobilenet_v1_int8_b64.json 2> $O/bench_mb64.err
tail -1 $O/bench_mobilenet_v1_int8_b64.json | cut -c1-400
cp $P $O/tbl_plan.txt
cd $R; TAMD_PLAN_CACHE=$O/tbl_plan.txt python tools/profile_layers.py mobilenet_v1 64 10 int8 2>&1 | grep -v "^Tengine" > $O/layers_mobilenet_v1_int8_b64.txt
tail -1 $O/layers_mobilenet_v1_int8_b64.txt
rm -rf $O/m_FETCH_SIZE $O/m_WRITE_SIZE $O/m_mfma $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE $O/pmc_plan.txt $O/tbl_plan.txt
find $O -name "*.db" -delete
