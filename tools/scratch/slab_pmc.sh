#!/bin/bash
ROOT=$(pwd); OUT=gpurun_out/slabpmc; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
for w in cull_slab cull_all_test; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $ROOT/$OUT/${w}_sq1 -o p -- python $ROOT/tools/run_workload.py --workload $w --steps 4 > $ROOT/$OUT/$w.sq1.log 2>&1 < /dev/null)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $ROOT/$OUT/${w}_sq2 -o p -- python $ROOT/tools/run_workload.py --workload $w --steps 4 > $ROOT/$OUT/$w.sq2.log 2>&1 < /dev/null)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/${w}_fetch -o p -- python $ROOT/tools/run_workload.py --workload $w --steps 4 > $ROOT/$OUT/$w.f.log 2>&1 < /dev/null)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/${w}_write -o p -- python $ROOT/tools/run_workload.py --workload $w --steps 4 > $ROOT/$OUT/$w.w.log 2>&1 < /dev/null)
done
python tools/pmc_summary.py $OUT/*_sq1 $OUT/*_sq2 $OUT/*_fetch $OUT/*_write > $OUT/summary.json 2> $OUT/summary.err
python - <<'PY'
import json
c=json.load(open('gpurun_out/slabpmc/summary.json'))
for k,v in sorted(c.items()):
    for kn,cv in v["counters_mean_per_launch"].items():
        if 'k_cull_tile' in kn: print(k, {a:round(b) for a,b in cv.items()})
PY
