# SQ counters of the several-frusta all-test launch (product, or LMX_LIB_PATH=variant through the environment of the call)
#   bash tools/gpu_call.sh cull8_counters [variant]
V=${1:-}
if [ -n "$V" ]; then export LMX_LIB_PATH=$ROOT/tools/_build/variants/$V/liblumix_mi355.so; fi
sq cull8_all_test
pmc cull8_all_test_mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS -- $W --workload cull8_all_test --steps 4
pmc cull8_all_test_mem SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM -- $W --workload cull8_all_test --steps 4
pmc_summary
prof cull8_all_test $W --workload cull8_all_test --steps 20
