#!/bin/bash
# Next round's first GPU call: A/B of the cull experiment knobs that were verified bit-exact on the simulated device at the end of round 3
# (tests/hostsim; LMX_HOSTSIM_EXTRA=<the same -D flags> python -m pytest tests/test_gpu_cull.py -m gpu --hostsim) but never timed.
#   base        the default build
#   hdr_ahead   -DLMX_CULL_HDR_AHEAD=1: the header -> cell -> class chain of all 8 chunks of a wave before the first group's loads (44 VGPRs, 84 SGPRs)
# Usage (on the GPU box, from the repo root):  bash tools/scratch/r04_ab.sh        -> gpurun_out/cull_ab_*.json + a table on stdout (~1 min)
cd "$(dirname "$0")/../.." || exit 1
python tools/build_variant.py base cull_kernels.hip "" > /dev/null
python tools/build_variant.py hdr_ahead cull_kernels.hip "-DLMX_CULL_HDR_AHEAD=1" > /dev/null
bash tools/scratch/cull_ab.sh base hdr_ahead base hdr_ahead
# k_keys_mesh (94 us per 1.05 M visible; footprint 231 B per visible entity by the traffic model AND by the PMC counters, 87 algorithmic). Three steps,
# all bit-exact on the simulated device (tests/test_sort_keys.py: removals / moves / re-sorts in every mode), none timed yet:
#   keys_split      LMX_KEYS_OPT_SPLIT_STATE = 1: lod / Pose::frame of the sorted set in a dense 8-byte-per-slot array       footprint 186 B, write sector use 0.26
#   keys_soa        ... = 2: the whole mirror as a structure of arrays (42 B per slot)                                       footprint 164 B, record loads 0.22 -> 0.70
#   keys_soa_stage  + LMX_KEYS_STAGE_PAIRS: the tile's pairs / records leave through LDS in position order                   write sector use 0.72
# The option's initial value is a compile-time default of the context: the variants re-compile lmx_capi_ctx.hip (where LmxContext is constructed).
python tools/build_variant.py keys_split lmx_capi_ctx.hip "-DLMX_KEYS_SPLIT_STATE_DEFAULT=1" > /dev/null
python tools/build_variant.py keys_soa lmx_capi_ctx.hip "-DLMX_KEYS_SPLIT_STATE_DEFAULT=2" > /dev/null
python tools/build_variant.py keys_soa_stage lmx_capi_ctx.hip,keys_kernels.hip "-DLMX_KEYS_SPLIT_STATE_DEFAULT=2 -DLMX_KEYS_STAGE_PAIRS=1" > /dev/null
bash tools/scratch/keys_ab.sh base keys_split keys_soa keys_soa_stage base keys_split keys_soa keys_soa_stage
# k_pose_palette: output through LDS staging rows (LMX_POSE_STAGE_OUT=1: palette rows, 2: + the absolute pose written back). The traffic model
# counts 0.41 -> 0.86 -> 1.00 sector use for the kernel's writes (73 % of its bytes); LDS per block 29.9 -> 42.4 -> 49.9 KiB (5 -> 3 blocks per
# CU), so this one can go either way. Kernel time of the skin workload (2000 instances x 64 bones) under rocprofv3.
ROOT=$(pwd); export TMPDIR=/tmp
python tools/build_variant.py pose_stage1 skin_kernels.hip "-DLMX_POSE_STAGE_OUT=1" > /dev/null
python tools/build_variant.py pose_stage2 skin_kernels.hip "-DLMX_POSE_STAGE_OUT=2" > /dev/null
for v in base pose_stage1 pose_stage2 base pose_stage1 pose_stage2; do
  OUT=gpurun_out/poseab_$v; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && LMX_LIB_PATH=$ROOT/tools/_build/variants/$v/liblumix_mi355.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT -o p -- python $ROOT/tools/run_workload.py --workload skin --steps 20 > $ROOT/$OUT/log.txt 2>&1 < /dev/null)
  python - "$v" <<'PY'
import csv, sys
v = sys.argv[1]
for r in csv.DictReader(open(f"gpurun_out/poseab_{v}/p_kernel_stats.csv")):
    if "k_pose_palette" in r["Name"] or "k_skin_shared" in r["Name"]:
        print("%-12s %-40s calls %s avg %.1f us min %.1f" % (v, r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
