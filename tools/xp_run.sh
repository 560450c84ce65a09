mkdir -p gpurun_out/r05l
for v in 0 1; do DSS2_WGRAD_XP=$v bash tools/prof_bench.sh > gpurun_out/r05l/prof_xp$v.txt 2>&1; cp gpurun_out/prof_bench_kernel_stats.csv gpurun_out/r05l/kernel_stats_xp$v.csv; done
cat gpurun_out/r05l/prof_xp0.txt; echo ----; cat gpurun_out/r05l/prof_xp1.txt
