#!/bin/bash
# Round 4, GPU call 31: clocks and package power while the cull kernels run back to back, and under a bare fill / copy (tools/write_probe.hip)
OUT=gpurun_out/r04; mkdir -p $OUT
smi() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; }
{
timeout 300 python tools/scratch/power_sample.py 2>&1 | grep -v amdgpu.ids
echo "== bare fill / copy loops (tools/write_probe.hip)"
./tools/_build/write_probe > $OUT/.wp.txt 2>&1 &
PID=$!
sleep 0.4; for k in 1 2 3 4; do smi; sleep 0.25; done
wait $PID; grep -E "fill float4 nt|copy float4|hipMemsetAsync" $OUT/.wp.txt | head -6
} 2>&1 | tee $OUT/cull_clocks_and_power.txt
rm -f $OUT/.wp.txt
