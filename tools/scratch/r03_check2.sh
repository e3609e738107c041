#!/bin/bash
# second round-3 check on the GPU box: pose kernel parity + time + LDS conflicts, the weak run's config-4 extra, event timing of the roofline leg
OUT=gpurun_out/r03b
ROOT=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_world_skin.py tests/test_gpu_bridges.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -n 3 $OUT/tests.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/target -o p -- python $ROOT/tools/run_workload.py --workload target --steps 12 > $ROOT/$OUT/target.log 2>&1 < /dev/null)
cp $OUT/target/p_kernel_stats.csv $OUT/target_kernel_stats.csv; rm -rf $OUT/target
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $ROOT/$OUT/skin_sq2 -o p -- python $ROOT/tools/run_workload.py --workload skin --steps 4 > $ROOT/$OUT/skin_sq2.log 2>&1 < /dev/null)
python tools/pmc_summary.py $OUT/skin_sq2 > $OUT/skin_sq2_counters.json 2> $OUT/skin_sq2.err; rm -rf $OUT/skin_sq2
python bench.py --no-extras --no-cpu-baseline --no-live-traffic > $OUT/bench_noextras.json 2> $OUT/bench_noextras.err
python bench.py --force-collective --no-cpu-baseline --no-live-traffic --big-entities 0 > $OUT/bench_fc_weak_full.json 2> $OUT/bench_fc_weak_full.err
grep -h "^\[rank 0\] config 4\|Traceback\|Error" $OUT/*.err | head
ls -la $OUT
