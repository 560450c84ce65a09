#!/bin/bash
# GPU: kernel timeline of one replayed training step of the reference driver's model line (tools/driverline_trace.py) at
# B = 64 and B = 4096 -> gpurun_out/trace_B<B>_post.txt.  Run from the repo root on the GPU box.
R="${GRAFT_REPO_ROOT:-$PWD}"
cd /tmp && export TMPDIR=/tmp
for B in 64 4096; do
  rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_B$B" -- python3 "$R/tools/driverline_trace.py" run $B > "$R/gpurun_out/trace_B$B.log" 2>&1
  f=$(find "$R/gpurun_out/trace_B$B" -name "*kernel_trace.csv" | head -1)
  python3 "$R/tools/driverline_trace.py" post "$f" > "$R/gpurun_out/trace_B${B}_post.txt" 2>&1
  rm -rf "$R/gpurun_out/trace_B$B"
done
