#!/bin/bash
# Round 4, GPU call 35: SQ / FETCH / WRITE counter passes over the north-star frame (k_skin_multi, k_pose_palette)
export TMPDIR=/tmp; OUT=gpurun_out/r04final; mkdir -p $OUT
bash tools/collect_counters.sh $OUT/target_counters "target" > $OUT/target_counters.log 2>&1
python tools/pmc_summary.py $OUT/target_counters/*_sq1 $OUT/target_counters/*_sq2 $OUT/target_counters/*_fetch $OUT/target_counters/*_write > $OUT/target_counters_summary.json
rm -rf $OUT/target_counters
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04final/target_counters_summary.json"))
for k,v in d.items():
    for kn,c in v.get("counters_mean_per_launch",{}).items():
        if "k_skin_multi" in kn or "k_pose_palette" in kn: print(k, kn, {a:round(b,1) for a,b in c.items()})
PY
