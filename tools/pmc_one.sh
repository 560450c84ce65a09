#!/bin/bash
# usage: tools/pmc_one.sh fwd|dgrad|wgrad|chain|wgrad2|wgrad3|stack64|stack4096   -> per-kernel PMC means (separate passes, no trace domains)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; W=${1:-fwd}
export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc1_${W}_$i -- python3 $R/tools/pmc_one.py $W 10 > $R/gpurun_out/pmc1_${W}_$i.log 2>&1
done
python3 - "$W" <<'PY'
import csv, glob, os, collections, sys
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); W=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for f in glob.glob(R+f"/gpurun_out/pmc1_{W}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "gemm_prop_kernel" in k or "gemm_chain_kernel" in k or "gemm_chain_sp_kernel" in k or "wgrad_kernel" in k or "wgrad16_kernel" in k or "wgrad16b_kernel" in k or "wgrad16t_kernel" in k or "wgrad16p_kernel" in k or "wgrad16q_kernel" in k or "wgrad16h_kernel" in k or "gemm_chain_sp6_kernel" in k or "reduce_slabs" in k or "stack_fwd_kernel" in k or "stack_bwd_kernel" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(R+f"/gpurun_out/pmc1_{W}_*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if k in agg: dur[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,d in agg.items():
    dd=sorted(dur[k]); print(k, f"launches={len(dd)} median_us={dd[len(dd)//2]:.1f}")
    for c,v in sorted(d.items()):
        print(f"   {c:30s} mean={sum(v)/len(v):16.1f}")
    g=d.get("GRBM_GUI_ACTIVE"); m=d.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if g and m:
        cyc=sum(g)/len(g)/8
        # (GRBM_GUI_ACTIVE / 8 over the launch's wall time is NOT a clock on dispatches this short -- it read 2.7-7.9 "GHz" in round 4;
        #  the in-kernel clock is s_memtime over wall time: bench.py's held_clock_ghz, tools/stamps.py chain)
        print(f"   -> kernel active cycles (GUI_ACTIVE/8) {cyc:.0f}; MFMA pipe busy {100*sum(m)/len(m)/(1024*cyc):.1f}% of SIMD-cycles")
PY
