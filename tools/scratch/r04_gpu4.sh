#!/bin/bash
# Round 4, GPU call 4: skin A/B (zero-weight skip) at 100 k instances; the default bench run with k_skin_multi as the default; the target frame under rocprofv3
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== skin A/B"; timeout 600 bash tools/scratch/skin_ab.sh > $OUT/skin_ab4.txt 2>&1; cat $OUT/skin_ab4.txt
echo "=== bench"; timeout 700 python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$? bytes=$(wc -c < $OUT/bench_line.json)"; cat $OUT/bench_line.json; cp bench_extra.json $OUT/ 2>/dev/null
grep -E "^\[extra (target|skin|keys_kernels|xform|pose|transform|config3)" $OUT/bench.err | cut -c1-330
echo "=== target frame under rocprofv3"; mkdir -p $OUT/target; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/target -o p -- python $ROOT/tools/run_workload.py --workload target --steps 6 > $ROOT/$OUT/target/log.txt 2>&1 < /dev/null); head -8 $OUT/target/p_kernel_stats.csv | cut -c1-200
