#!/bin/bash
# Round-end measurement set on the GPU box (run through gpurun): writes everything under gpurun_out/final/.
#   bench line (with CPU baseline), rocprofv3 kernel stats of the same program, PMC passes for the two dominant
#   kernels, the per-configuration table and the collate micro-benchmark.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
cd $R
python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/final_stats -o s -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
cp /tmp/final_stats/*kernel_stats.csv $O/kernel_stats_bench_c2.csv
bash tools/pmc_one.sh chain > $O/pmc_gemm_chain.txt 2>&1
bash tools/pmc_one.sh wgrad2 > $O/pmc_wgrad_batched.txt 2>&1
rm -rf $R/gpurun_out/pmc1_*
python3 tools/cfgbench.py > $O/cfgbench.txt 2>&1
python3 tools/collate_bench.py > $O/collate_bench.txt 2>&1
ls -la $O
