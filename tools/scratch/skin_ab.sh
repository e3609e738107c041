#!/bin/bash
# Round 4 A/B of the shared-mesh skinning kernels (tools/skin_probe.hip): <instances> x 10 000 vertices x 64 bones.
# mesh 0 = worst case (4 random bones of 64 per vertex), 1 = character-like (a tile touches ~27 bones, 1-2 influences).
#   skin_probe_t<threads>_pf<blocks ahead>: k_skin_multi with that block size and L2 touch distance (0 = none), I = 1, 2, 4 x 1 / 2 vertex ranges
#   arguments: instances, instances per k_skin_shared block, kind (0 k_skin_vertices, 1 k_skin_shared, 2 k_skin_multi), mesh, hot palettes
cd "$(dirname "$0")/../_build" || exit 1
for n in 20000 100000; do
 for mesh in 0 1; do
  echo "== $n instances, mesh=$mesh: k_skin_shared"; ./skin_probe_t512_pf0 $n 64 1 $mesh | grep -v "^tile"
  for p in t512_pf0 t512_pf768 t512_pf1536 t1024_pf512 t1024_pf0; do
    echo "== $n instances, mesh=$mesh: k_skin_multi $p"; ./skin_probe_$p $n 64 2 $mesh | grep -v "^tile"
  done
 done
done
echo "== 100000 instances, mesh=0, HOT palettes (every block reads instance 0's): k_skin_multi t512_pf0"; ./skin_probe_t512_pf0 100000 64 2 0 1 | grep -v "^tile"
