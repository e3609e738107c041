#!/bin/bash
# Round 4 A/B of the shared-mesh skinning kernels (tools/skin_probe.hip), 20 000 instances x 10 000 vertices x 64 bones.
# mesh 0 = worst case (4 random bones of 64 per vertex), 1 = character-like (a tile touches ~27 bones, 1-2 influences).
#   skin_probe_p<pipe>_t<threads>: k_skin_multi with that software-pipeline depth and block size, I = 1..16 x 1 / 2 / 4 vertex ranges
#   kind 1 = k_skin_shared (rounds 2 / 3), kind 0 = k_skin_vertices
cd "$(dirname "$0")/../_build" || exit 1
for mesh in 0 1; do
  echo "== k_skin_shared mesh=$mesh"; ./skin_probe_p2_t512 20000 64 1 $mesh | grep -v "^tile"
  for p in p2_t512 p2_t1024 p4_t1024; do
    echo "== k_skin_multi $p mesh=$mesh"; ./skin_probe_$p 20000 64 2 $mesh | grep -v "^tile" | grep -v "I= 8\|I=16"
  done
done
echo "== 100 000 instances (the target frame's count), mesh 0"; ./skin_probe_p2_t512 100000 64 2 0 | grep "I= [12] splits=1"; ./skin_probe_p2_t1024 100000 64 2 0 | grep "I= [12] splits=1"
