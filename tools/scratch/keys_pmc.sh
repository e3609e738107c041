#!/bin/bash
ROOT=$(pwd); OUT=gpurun_out/keyspmc; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/$OUT/keys_$c -o p -- python $ROOT/tools/run_workload.py --workload keys --steps 4 > $ROOT/$OUT/$c.log 2>&1 < /dev/null)
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $ROOT/$OUT/keys_sq1 -o p -- python $ROOT/tools/run_workload.py --workload keys --steps 4 > $ROOT/$OUT/sq1.log 2>&1 < /dev/null)
python tools/pmc_summary.py $OUT/keys_FETCH_SIZE $OUT/keys_WRITE_SIZE $OUT/keys_sq1 > $OUT/summary.json 2> $OUT/summary.err
python - <<'PY'
import json
c=json.load(open('gpurun_out/keyspmc/summary.json'))
for k,v in c.items():
    for kn,cv in v["counters_mean_per_launch"].items():
        if 'k_keys_mesh' in kn: print(k, cv)
PY
