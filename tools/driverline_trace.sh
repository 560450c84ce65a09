#!/bin/bash
# tools/driverline_trace.sh [B]  -> gpurun_out/driverline_trace_B<B>.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-64}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/dl_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/dl_trace -o t -- python3 $R/tools/driverline_trace.py run $B > $R/gpurun_out/driverline_trace_B$B.log 2>&1
f=$(find /tmp/dl_trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/driverline_trace.py post $f > $R/gpurun_out/driverline_trace_B$B.txt 2>&1
tail -45 $R/gpurun_out/driverline_trace_B$B.txt
tail -2 $R/gpurun_out/driverline_trace_B$B.log
