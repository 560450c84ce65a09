#!/bin/bash
# rocprofv3 kernel stats of tools/exp_c2.py under one setting: tools/prof_exp.sh <tag> [exp_c2 args...] -> gpurun_out/prof_<tag>.csv (+ top rows on stdout)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o s -- python3 $R/tools/exp_c2.py "$@" > $R/gpurun_out/prof_$tag.log 2>&1
find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_$tag.csv \;
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_$tag.csv")))
rows=[r for r in rows if int(r['Calls'])>100]
tot=sum(float(r['TotalDurationNs'])/int(r['Calls']) for r in rows)
print("$tag: sum of per-step kernel averages %.1f us" % (tot/1e3))
for r in rows[:18]:
    print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
