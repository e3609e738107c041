#!/bin/bash
# Round 4 A/B of the shared-mesh skinning kernels (tools/skin_probe.hip): <instances> x 10 000 vertices x 64 bones.
# mesh 0 = worst case (4 random bones of 64 per vertex), 1 = character-like (a tile touches ~27 bones, 1-2 influences).
#   skin_probe_pf<blocks ahead>_skip<0|1>: k_skin_multi (512 threads) with that L2 touch distance and with / without the wave-uniform skip of zero-weight bone slots
#   arguments: instances, instances per k_skin_shared block, kind (0 k_skin_vertices, 1 k_skin_shared, 2 k_skin_multi), mesh, hot palettes
cd "$(dirname "$0")/../_build" || exit 1
for n in 100000; do
 for mesh in 0 1; do
  echo "== $n instances, mesh=$mesh: k_skin_shared"; ./skin_probe_pf768_skip1 $n 64 1 $mesh | grep -v "^tile"
  for p in pf768_skip1 pf768_skip0; do
    echo "== $n instances, mesh=$mesh: k_skin_multi $p"; ./skin_probe_$p $n 64 2 $mesh | grep -v "^tile" | grep "splits=1" | grep -v "I= 4"
  done
 done
done
