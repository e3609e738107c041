#!/bin/bash
# Round 4, GPU call 25: is k_skin_multi slower per vertex at 100 000 instances than at 20 000 because the launch is LONG (sustained clocks / power) or because the
# output is LARGE? 20 000 instances launched 1 / 5 / 20 times back to back per timed region; 100 000 once and 4 times
OUT=gpurun_out/r04; mkdir -p $OUT
{
for r in 1 5 20; do echo "== 20000 instances x $r launches back to back"; PROBE_REPEAT=$r ./tools/_build/skin_probe_base 20000 2 2 0 0 64 | grep "I= 2 splits=1"; done
for r in 1 4; do echo "== 100000 instances x $r launches back to back"; PROBE_REPEAT=$r ./tools/_build/skin_probe_base 100000 2 2 0 0 64 | grep "I= 2 splits=1"; done
} 2>&1 | tee $OUT/skin_long_launch_or_large_output.txt
