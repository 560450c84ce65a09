#!/bin/bash
# rocprofv3 kernel stats of bench.py (C2) -> gpurun_out/prof_bench_kernel_stats.csv
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o s -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --min-window-seconds 0.5 > $R/gpurun_out/prof_bench.log 2>&1
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_bench_kernel_stats.csv \;
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_bench_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
