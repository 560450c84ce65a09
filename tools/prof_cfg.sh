#!/bin/bash
# rocprofv3 kernel stats of one cfgbench configuration (eager mode): tools/prof_cfg.sh <tag> <cfgbench selector...>
# writes gpurun_out/prof_<tag>_kernel_stats.csv
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
export TMPDIR=/tmp CFG_GRAPH=0
cd /tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o s -- python3 $R/tools/cfgbench.py "$@" > $R/gpurun_out/prof_$tag.log 2>&1
cp /tmp/prof_$tag/*kernel_stats.csv $R/gpurun_out/prof_${tag}_kernel_stats.csv 2>/dev/null || find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_${tag}_kernel_stats.csv \;
head -14 $R/gpurun_out/prof_${tag}_kernel_stats.csv | cut -c1-200
