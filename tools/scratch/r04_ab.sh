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
# k_keys_mesh: lod / Pose::frame of the sorted set in a dense per-slot array (LMX_KEYS_OPT_SPLIT_STATE; bit-exact on the simulated device,
# the traffic model of tools/traffic_model.py says 231 -> 186 B of footprint per visible entity). The option's initial value is a
# compile-time default of the context: the variant re-compiles lmx_capi_ctx.hip (where LmxContext is constructed) with it switched on.
python tools/build_variant.py keys_split lmx_capi_ctx.hip "-DLMX_KEYS_SPLIT_STATE_DEFAULT=1" > /dev/null
bash tools/scratch/keys_ab.sh base keys_split base keys_split
