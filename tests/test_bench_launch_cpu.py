"""CPU: `python3 bench.py --gpus N` with no launcher environment must start N ranks by itself (torch.distributed.run
children spawned before the parent touches the GPU), relay rank 0's JSON line and return the children's exit code.
--dry-run walks exactly that path with gloo and no GPU call (missing #2)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_gpus2_without_a_launcher_spawns_two_ranks():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout            # ONE JSON line, from rank 0
    res = json.loads(lines[0])
    assert res == {"dry_run": True, "n_gpus": 2, "ranks_seen": 3, "steps": 3, "warmup": 1}


def test_bench_dry_run_single_process():
    p = _run(["--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_under_a_launcher_with_the_wrong_world_size_fails():
    p = _run(["--gpus", "2", "--dry-run"], env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert p.returncode != 0


def test_a_hung_capture_leg_at_world_2_leaves_the_eager_line_last_and_exits_0():
    """at world > 1 rank 0 prints the eager line BEFORE the hipGraph leg; if that leg hangs (a collective
    that never completes inside a capture) every rank's watchdog ends its process with rc 0 -- the first 8-GPU run cannot
    lose its measurement.  --simulate-hung-capture blocks every rank exactly there."""
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--simulate-hung-capture", "--graph-timeout", "3"])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[-1])
    assert res["n_gpus"] == 2 and res["config"]["mode"] == "eager" and "partial" not in res
    assert "no answer within 3 s" in p.stderr


def test_a_hung_capture_leg_at_world_1_prints_the_eager_line_from_the_watchdog():
    p = _run(["--dry-run", "--simulate-hung-capture", "--graph-timeout", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 1 and "hung" in res["partial"] and "no answer" in res["config"]["hipgraph"]
