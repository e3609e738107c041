#!/bin/bash
# Round 4, GPU call 30: clocks and power while k_skin_multi runs back to back (is the kernel held by the power limit? its three streams - stores, LDS reads,
# VALU - add up instead of overlapping), worst-case and character-like mesh, against the idle chip
OUT=gpurun_out/r04; mkdir -p $OUT
smi() { rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|GPU use" | tr '\n' ';' | cut -c1-400; echo; }
{
echo "== idle"; smi
for mesh in 0 1; do
  echo "== k_skin_multi, mesh $mesh, 100000 instances, 150 launches back to back per timed region"
  PROBE_REPEAT=150 ./tools/_build/skin_probe_base 100000 2 2 $mesh 0 64 > $OUT/.skin_bg.txt 2>&1 &
  PID=$!
  sleep 2.5; for k in 1 2 3; do smi; sleep 0.4; done
  wait $PID; grep "I= 2 splits=1" $OUT/.skin_bg.txt
done
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
} 2>&1 | tee $OUT/skin_clocks_and_power.txt
rm -f $OUT/.skin_bg.txt
