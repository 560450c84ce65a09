mkdir -p gpurun_out/r05q
python -m pytest tests/test_gpu_plan.py -x -q -m gpu > gpurun_out/r05q/plan.log 2>&1; tail -25 gpurun_out/r05q/plan.log | cut -c1-220
python bench.py --no-cpu-baseline --min-window-seconds 1.5 --ramp-seconds 1 --other-seconds 0.7 > gpurun_out/r05q/bench.json 2> gpurun_out/r05q/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05q/bench.json'))
print(d['ms_per_step'], d['config']['ms_per_step_by_mode'])
for k,v in d.get('other_configs',{}).items():
    print(k[:40], {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk.startswith('ms_per') or kk in ('plan_launches','plan_error','skipped','replay_error')})
PY
tail -3 gpurun_out/r05q/bench.err
