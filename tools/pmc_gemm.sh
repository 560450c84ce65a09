#!/bin/bash
# PMC counters for the dominant kernel (separate passes; no trace domains mixed in)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/kbench.py --reps 3 --rounds 1 > $R/gpurun_out/pmc_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+"/gpurun_out/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "gemm_prop_kernel<2, 3>" in k or "wgrad_kernel<2, 3, 4>" in k:
            agg[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()):
        print(f"   {c:34s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
