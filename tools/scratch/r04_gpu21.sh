#!/bin/bash
# Round 4, GPU call 21: instruction counts of k_cull_tile<F = 0> on the 8-frusta all-test launch, whole and with phases compiled out (LMX_CULL8_PROBE)
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
for v in base cull8_probe4 cull8_probe6; do
  LIB=$ROOT/tools/_build/variants/$v/liblumix_mi355.so; [ $v = base ] && LIB=$ROOT/lumixengine_amd/liblumix_mi355.so
  D=gpurun_out/cull8pmc_$v; rm -rf $D
  (cd /tmp && LMX_LIB_PATH=$LIB timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $ROOT/$D -o p -- python $ROOT/tools/run_workload.py --workload cull8_all_test --steps 4 > $ROOT/$D.log 2>&1 < /dev/null)
  python tools/pmc_summary.py $D | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    for kn,c in v.get('counters_mean_per_launch',{}).items():
        if 'k_cull_tile' in kn: print('$v', kn, json.dumps(c))
"
done | tee $OUT/cull8_probe_counters.txt
