#!/bin/bash
# A/B of library build variants through bench.py's headline + roofline legs (no extras)
for v in "$@"; do
  LMX_LIB_PATH=tools/_build/variants/$v/liblumix_mi355.so python bench.py --no-extras --no-cpu-baseline --no-live-traffic > gpurun_out/bench_ab_$v.json 2> gpurun_out/bench_ab_$v.err
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
b=json.load(open(f"gpurun_out/bench_ab_{v}.json"))
l=b["roofline"]["legs"]
print("%-6s step %.2f us cull-only %.2f us | default warm %.2f cold %.2f | accept %.1f/%.1f | all_test %.2f/%.2f | slab %.2f/%.2f us" % (v, b["ms_per_step"]*1e3, b["ms_per_step_cull_only"]*1e3,
  l["default_camera"]["warm_avg_launch_ms"]*1e3, l["default_camera"]["cold_avg_launch_ms"]*1e3, l["all_accept"]["warm_avg_launch_ms"]*1e3, l["all_accept"]["cold_avg_launch_ms"]*1e3,
  l["all_test"]["warm_avg_launch_ms"]*1e3, l["all_test"]["cold_avg_launch_ms"]*1e3, l["all_cell_test_normal_radii"]["warm_avg_launch_ms"]*1e3, l["all_cell_test_normal_radii"]["cold_avg_launch_ms"]*1e3))
PY
done
