mkdir -p gpurun_out/r05j
DSS2_WGRAD_XP=1 python -m pytest tests/test_gpu_xplanes.py -q -m gpu 2>&1 | tail -8 > gpurun_out/r05j/xplanes_q.log
python tools/xp_bench.py --only-wgrad --no-rs2 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r05j/xp_q_nors2.txt
python tools/xp_bench.py --only-wgrad 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r05j/xp_q_rs2.txt
for l in qb18 qb12; do DSS2_LIB=$PWD/deep-statistical-solver-for-distribution-system-state-estimation_amd/libdss2_$l.so python tools/xp_bench.py --only-wgrad 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r05j/xp_${l}_rs2.txt; done
DSS2_LIB=$PWD/deep-statistical-solver-for-distribution-system-state-estimation_amd/libdss2_qst.so python tools/qstamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05j/qstamps.txt
B="--no-cpu-baseline --no-other-configs --min-window-seconds 1.5 --ramp-seconds 1"
DSS2_WGRAD_XP=0 python bench.py $B > gpurun_out/r05j/bench_xp0.json 2>/dev/null
DSS2_WGRAD_XP=1 python bench.py $B > gpurun_out/r05j/bench_xp1.json 2>/dev/null
cat gpurun_out/r05j/xplanes_q.log; for f in gpurun_out/r05j/xp_*.txt; do echo $f; cat $f; done; cat gpurun_out/r05j/qstamps.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05j/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],4), d['config']['ms_per_step_by_mode'], 'chain', round(d['roofline']['avg_launch_us'],1))
    except Exception as e: print(f, 'ERR', e)
PY
