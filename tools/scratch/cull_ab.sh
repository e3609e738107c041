#!/bin/bash
# A/B of cull kernel build variants (tools/build_variant.py): tile variants 1 (streaming) and 4 (latency), warm + cold kernel time
cd "$(dirname "$0")/../.." || exit 1
for v in "$@"; do
  LMX_LIB_PATH=tools/_build/variants/$v/liblumix_mi355.so python tools/cull_sweep.py --variants 1,4 --no-coldw --reps 40 --tag $v --out gpurun_out/cull_ab_$v.json 2>&1 | grep '^{' | python -c '
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print("%-7s %-8s %-11s v%d  vis %8d  warm %7.2f us  cold %7.2f us  wall %7.2f us" % (r["tag"], r["scene"], r["leg"], r["variant"], r["visible"], r["warm_kernel_us"], r["cold_kernel_us"], r["wall_us"]))'
done
