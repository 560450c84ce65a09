#!/bin/bash
# rocprofv3 kernel timeline of one bench.py step in distributed mode at world size 1 (RANK=0 WORLD_SIZE=1): shows the host-side gaps the
# two per-step collectives open in an eager step (tools/host_rate.py has the host-vs-GPU time per step).
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
cd /tmp; rm -rf /tmp/dtrace
rocprofv3 --kernel-trace --output-format csv -d /tmp/dtrace -o t -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/dist_trace.log 2>&1
f=$(find /tmp/dtrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last full step: between the last two wls_partials kernels
idx = [i for i, n in enumerate(names) if "wls_partials" in n]
a, b = idx[-3], idx[-2]
prev = int(rows[a - 1]["End_Timestamp"])
tot_gap = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev
    nm = r["Kernel_Name"].split("(")[0][-70:]
    print(f"{nm:70s} dur {1e-3*(e-s):7.1f} gap {1e-3*gap:7.1f}")
    tot_gap += max(gap, 0); prev = max(prev, e)
print("step span us", 1e-3 * (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])), "gaps", 1e-3 * tot_gap)
PY
tail -1 $R/gpurun_out/dist_trace.log | cut -c1-200
