#!/bin/bash
# A/B of k_skin_shared / k_skin_vertices build variants (tools/skin_probe.hip), 20 000 instances x 10 000 vertices
# mesh 0 = worst case (4 random bones of 64 per vertex), 1 = character-like (a tile touches ~27 bones, 1-2 influences)
cd "$(dirname "$0")/../_build" || exit 1
for mesh in 0 1; do
  for p in r02 dma0 dma14 dma13 dma12 dma23 dma11 dma12_m8; do
    [ -x ./skin_probe_$p ] || continue
    echo "== skin_probe_$p shared mesh=$mesh"
    ./skin_probe_$p 20000 64 1 $mesh | grep -v "^tile"
  done
done
for p in r02 dma12; do
  echo "== skin_probe_$p streaming (k_skin_vertices), 2000 instances"
  ./skin_probe_$p 2000 64 0 0 | grep -v "^tile"
done
