# run a probe binary from tools/_build:  bash tools/gpu_call.sh probe <name> [args]
P=$1; shift
timeout 300 ./tools/_build/$P "$@" 2>&1 | tee "$OUT/$P.txt"
