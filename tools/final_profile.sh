#!/bin/bash
# Round-end measurement set on the GPU box (run through gpurun): writes everything under gpurun_out/final/.
#   bench line (with the CPU baseline sweep), rocprofv3 kernel stats of the same program, PMC passes for the dominant
#   kernels and for the cache-busting scatter-add, the per-configuration table, the driver's model line, batch assembly.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
cd $R
python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/final_stats -o s -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-routes --min-window-seconds 0.5 > $O/bench_under_rocprof.log 2>&1
find /tmp/final_stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bench_c2.csv \;
cd $R
bash tools/pmc_one.sh chain > $O/pmc_gemm_chain.txt 2>&1
bash tools/pmc_one.sh wgrad3 > $O/pmc_wgrad_batched.txt 2>&1
bash tools/pmc_one.sh stack4096 > $O/pmc_stack_B4096.txt 2>&1
bash tools/pmc_one.sh stack64 > $O/pmc_stack_B64.txt 2>&1
rm -rf $R/gpurun_out/pmc1_*
# scatter-add, cache-busting working set: FETCH_SIZE and WRITE_SIZE in separate passes (no trace domains besides kernel-trace)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_scatter_$c -- python3 $R/tools/pmc_scatter.py 32768 10 > /dev/null 2>&1
done
python3 - > $O/pmc_scatter_add_b32768.txt <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list); dur = []
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pmc_scatter_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "segment_sum_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"/tmp/pmc_scatter_{c}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "segment_sum_kernel" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
dur.sort()
print("dss2::segment_sum_kernel, B = 32768 CIGRE-14 graphs, H = 128 (msg 470 MB, out 252 MB, indices 5.6 MB)")
print(f"launches {len(dur)}  median {dur[len(dur)//2]:.1f} us (under the counter passes)")
f = sum(agg["FETCH_SIZE"]) / max(len(agg["FETCH_SIZE"]), 1); w = sum(agg["WRITE_SIZE"]) / max(len(agg["WRITE_SIZE"]), 1)
print(f"FETCH_SIZE mean {f:.1f} KB (x2 on gfx950 for wide streaming reads = {2*f/1e3:.1f} MB)   WRITE_SIZE mean {w:.1f} KB = {w/1e3:.1f} MB")
alg = 4.0 * 917504 * 128 + 4.0 * 491520 * 128 + 4.0 * 917504 + 4.0 * 491521
print(f"HBM bytes per launch (2 x FETCH + WRITE) = {(2*f + w)*1e3/1e6:.1f} MB vs algorithmic {alg/1e6:.1f} MB")
PY
cd $R
python3 tools/cfgbench.py > $O/cfgbench.txt 2>&1
# kernel stats of the tall-tile configurations (eager mode) and the counters of their weight-gradient launch
bash tools/prof_cfg.sh c3 "C3 ober_sub" > /dev/null 2>&1; cp $R/gpurun_out/prof_c3_kernel_stats.csv $O/kernel_stats_c3_ober_sub.csv 2>/dev/null
bash tools/prof_cfg.sh c3p "ober179" > /dev/null 2>&1; cp $R/gpurun_out/prof_c3p_kernel_stats.csv $O/kernel_stats_c3_ober179.csv 2>/dev/null
PMC_GRID=ober_sub PMC_B=1024 bash tools/pmc_one.sh wgrad3 > $O/pmc_wgrad_tall_c3.txt 2>&1
rm -rf $R/gpurun_out/pmc1_*
python3 tools/driverline.py > $O/driverline.txt 2>&1
python3 tools/collate_bench.py > $O/collate_bench.txt 2>&1
# round 6: fresh mixed-topology batches (C5), non-finite semantics, power / clock under the step, the driver line on ober_sub, NaN bit patterns
python3 tools/c5_fresh_probe.py 2>&1 | grep -v amdgpu > $O/c5_fresh_probe.txt
python3 tools/nonfinite_probe.py 2>&1 | grep -v amdgpu > $O/nonfinite_probe.txt
python3 tools/power_probe.py 2>&1 | grep -v amdgpu > $O/power_probe.txt
python3 tools/ober_driverline.py 64 256 512 1024 2>&1 | grep "SkipPFN driver" > $O/ober_driverline.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-result tools/micro/nan_bits.hip -o /tmp/nan_bits 2>/dev/null && /tmp/nan_bits > $O/nan_bits.txt 2>&1
cp $R/gpurun_out/parity_errors.jsonl $O/ 2>/dev/null
python3 tools/chainbench.py > $O/chainbench.txt 2>&1
CHAINBENCH_BF16=0 CHAINBENCH_STEP=0 python3 tools/chainbench.py > $O/chainbench_fp32_mfma.txt 2>&1
python3 tools/accuracy_bf16x6.py > $O/accuracy_bf16x6.txt 2>&1
bash tools/driverline_trace.sh 64 > /dev/null 2>&1; cp $R/gpurun_out/driverline_trace_B64.txt $O/ 2>/dev/null
bash tools/driverline_trace.sh 4096 > /dev/null 2>&1; cp $R/gpurun_out/driverline_trace_B4096.txt $O/ 2>/dev/null
# phase stamps of the whole-stack kernels (needs the diagnostic build tools/diag_lib/libdss2_sstamps.so, see tools/stamps.py)
P=$R/deep-statistical-solver-for-distribution-system-state-estimation_amd
if [ -f tools/diag_lib/libdss2_sstamps.so ]; then
  DSS2_LIB=tools/diag_lib/libdss2_sstamps.so python3 tools/stamps.py stack 64 2>&1 | grep -v amdgpu > $O/stack_stamps_B64.txt
  DSS2_LIB=tools/diag_lib/libdss2_sstamps.so python3 tools/stamps.py stack 4096 2>&1 | grep -v amdgpu > $O/stack_stamps_B4096.txt
fi
# phase stamps of the split-plane layer chain (needs tools/diag_lib/libdss2_cstamps.so, see tools/stamps.py)
if [ -f tools/diag_lib/libdss2_cstamps.so ]; then
  DSS2_LIB=tools/diag_lib/libdss2_cstamps.so python3 tools/stamps.py chain 1024 2>&1 | grep -v amdgpu > $O/chain_stamps_B1024.txt
  DSS2_LIB=tools/diag_lib/libdss2_cstamps.so python3 tools/stamps.py chain 4096 2>&1 | grep -v amdgpu > $O/chain_stamps_B4096.txt
  STAMPS_F16=0 DSS2_LIB=tools/diag_lib/libdss2_cstamps.so python3 tools/stamps.py chain 4096 2>&1 | grep -v amdgpu > $O/chain_stamps_B4096_bf16x6.txt
fi
# phase stamps of the f16x3 weight gradient (needs tools/diag_lib/libdss2_hstamps.so: -DDSS2_STAMPS)
if [ -f tools/diag_lib/libdss2_hstamps.so ]; then
  DSS2_LIB=tools/diag_lib/libdss2_hstamps.so python3 tools/stamps.py wgradh 2>&1 | grep -v amdgpu > $O/wgrad16h_stamps_C2.txt
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/micro/f16x3_probe.hip -o /tmp/f16x3 2>/dev/null && /tmp/f16x3 > $O/f16x3_probe.txt 2>&1
# the packed-fp32 reproducer (tools/micro/pkfma_beside_mfma.hip) and the 200-launch stress of the real kernel
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/pkfma_beside_mfma.hip -o /tmp/pkfma 2>/dev/null && /tmp/pkfma > $O/pkfma_micro.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/work_beside_mfma.hip -o /tmp/wbm 2>/dev/null && /tmp/wbm > $O/work_beside_mfma.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/micro/mfma_shape_clock.hip -o /tmp/msc 2>/dev/null && /tmp/msc > $O/mfma_shape_clock.txt 2>&1
python3 tools/pk_stress.py 200 2>&1 | grep -v amdgpu > $O/pk_stress_shipped.txt
ls -la $O
