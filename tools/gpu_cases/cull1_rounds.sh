# 1-frustum all-test launch by entity count: does a grid that is a whole number of rounds of resident blocks (2048 x 2048-sphere tiles = 4.19 M entities per round) run nearer the copy rate than BASELINE's 10 M (2.38 rounds)?
for n in 8388608 10000000 12582912 16777216; do
	prof n$n $W --workload cull_all_test --steps 30 --cold read --entities $n
	echo "$n entities: $(grep 'k_cull_tile<1' "$OUT/n${n}_kernel_stats.csv" | awk -F, '{print "calls", $(NF-6), "avg ns", $(NF-4), "min ns", $(NF-2)}' | head -1)"
done 2>&1 | tee "$OUT/cull1_rounds.txt"
