#!/bin/bash
for v in "$@"; do
  LMX_LIB_PATH=tools/_build/variants/$v/liblumix_mi355.so python bench.py --headline-only --no-extras --no-cpu-baseline --steps 2000 2> /dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read()); print('$v step %.2f us  cull only %.2f us' % (b['ms_per_step']*1e3, b['ms_per_step_cull_only']*1e3))"
done
python -m pytest tests/test_gpu_adapter.py tests/test_gpu_exchange.py -m gpu -x -q > gpurun_out/t49.log 2>&1; grep -h "passed\|failed" gpurun_out/t49.log
