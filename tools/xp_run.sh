mkdir -p gpurun_out/r05n
python -m pytest tests/test_gpu_bf16x6.py tests/test_gpu_xplanes.py tests/test_gpu_rccl.py -x -q -m gpu > gpurun_out/r05n/gpu_part.log 2>&1; tail -3 gpurun_out/r05n/gpu_part.log
B="--no-cpu-baseline --no-other-configs --min-window-seconds 2 --ramp-seconds 1.5"
for v in 0 1 0 1; do DSS2_WGRAD_RANGES=$v python bench.py $B > gpurun_out/r05n/bench_rg${v}_$RANDOM.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05n/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],4), d['config']['ms_per_step_by_mode'], 'chain', round(d['roofline']['avg_launch_us'],1), 'wgrad', d.get('roofline_wgrad',{}).get('avg_launch_us'))
    except Exception as e: print(f, 'ERR', e)
PY
