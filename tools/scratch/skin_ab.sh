#!/bin/bash
# Round 4 A/B of the shared-mesh skinning kernels (tools/skin_probe.hip), 20 000 instances x 10 000 vertices x 64 bones.
# mesh 0 = worst case (4 random bones of 64 per vertex), 1 = character-like (a tile touches ~27 bones, 1-2 influences).
#   skin_probe_pipe{2,3,4}: k_skin_multi with that software-pipeline depth (-DLMX_MULTI_PIPE), I = 1..16 x 1 / 2 / 4 vertex ranges
#   kind 1 = k_skin_shared (rounds 2 / 3), kind 0 = k_skin_vertices
cd "$(dirname "$0")/../_build" || exit 1
for mesh in 0 1; do
  echo "== k_skin_shared mesh=$mesh"; ./skin_probe_pipe2 20000 64 1 $mesh | grep -v "^tile"
  for p in 2 3 4; do
    echo "== k_skin_multi pipe=$p mesh=$mesh"; ./skin_probe_pipe$p 20000 64 2 $mesh | grep -v "^tile"
  done
done
echo "== k_skin_vertices, 2000 instances"; ./skin_probe_pipe2 2000 64 0 0 | grep -v "^tile"
