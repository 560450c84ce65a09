mkdir -p gpurun_out/r05r
python -m pytest tests -x -q -m gpu > gpurun_out/r05r/gpu_suite.log 2>&1; tail -5 gpurun_out/r05r/gpu_suite.log | cut -c1-300
python bench.py --no-cpu-baseline --min-window-seconds 2 --ramp-seconds 1.5 --other-seconds 0.7 > gpurun_out/r05r/bench.json 2> gpurun_out/r05r/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05r/bench.json'))
print(d['value'], d['ms_per_step'], d['config']['mode'], d['config']['ms_per_step_by_mode'])
r=d['roofline']; print({k:r[k] for k in ('frac','avg_launch_us','held_clock_ghz','frac_at_held_clock','head_flops_per_launch_included') if k in r})
print(d.get('roofline_wgrad',{}).get('avg_launch_us'), d.get('roofline_wgrad',{}).get('frac'))
for k,v in d.get('other_configs',{}).items():
    print(k[:40], {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk.startswith('ms_per') or kk in ('plan_launches','plan_error','skipped','replay_error')})
PY
tail -3 gpurun_out/r05r/bench.err
