# a subset of the GPU suite:  bash tools/gpu_call.sh pytest <pytest args...>
timeout 1500 python -m pytest "$@" -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"; tail -n 25 "$OUT/pytest.log"
