#!/usr/bin/env python3
"""Post-processing for `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...`: the kernels of the LAST
training step in launch order with their durations and grids (argv[1] = <dir>, argv[2] = how many kernels back)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n_k = int(sys.argv[2] if len(sys.argv) > 2 else 40)
if len(sys.argv) > 3:      # argv[3]: a kernel-name fragment; print n_k kernels from its LAST occurrence that has n_k successors
    idx = [i for i, r in enumerate(rows) if sys.argv[3] in r["Kernel_Name"] and i + n_k <= len(rows)]
    rows = rows[idx[-1]:idx[-1] + n_k]
else:
    rows = rows[-n_k:]
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-44:]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{n:46s} {d:8.1f} us   grid {r.get('Grid_Size_X', '')} x {r.get('Grid_Size_Y', '')} x {r.get('Grid_Size_Z', '')}  wg {r.get('Workgroup_Size_X', '')}")
