#!/bin/bash
# Round 4: what bounds k_skin_multi at 100 000 instances (sustained clocks)? I = 2, worst-case mesh; parts compiled out / knobs.
cd "$(dirname "$0")/../_build" || exit 1
for p in base pipe4 pipe6 plain_stores t1024 nostores nolds neither base; do
  echo "== $p"; ./skin_probe_$p 100000 64 2 0 | grep "I= 2 splits=1"
done
echo "== base, mesh 1"; ./skin_probe_base 100000 64 2 1 | grep "I= 2 splits=1"
echo "== nostores, mesh 1"; ./skin_probe_nostores 100000 64 2 1 | grep "I= 2 splits=1"
