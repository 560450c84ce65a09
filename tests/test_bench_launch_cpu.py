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


def test_a_hung_leg_at_world_2_is_recorded_in_the_last_line_and_in_the_exit_status():
    """VERDICT r4 #3 + r5 weak #9: at world > 1 rank 0 prints the eager line BEFORE the replay legs; if a leg hangs (a collective
    that never completes inside a capture) every rank's watchdog ends its process -- rank 0 first prints the FINAL line, which carries
    the eager measurement AND the hang record, and the run's exit status is non-zero (3): the measurement survives, the hang is not
    mistaken for a clean run.  --simulate-hung-capture blocks every rank exactly there."""
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--simulate-hung-capture", "--graph-timeout", "3"])
    assert p.returncode != 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 2, p.stdout
    assert "partial" in json.loads(lines[0])
    res = json.loads(lines[-1])
    assert res["n_gpus"] == 2 and res["config"]["mode"] == "eager" and "partial" not in res
    assert res["hang"]["leg"] == "hipGraph replay" and res["hang"]["exit_status"] == 3 and res["legs"] == {"eager": "ok", "hipGraph replay": "hung"}
    assert "no answer within 3 s" in p.stderr


def test_a_hung_leg_at_world_1_prints_the_final_line_from_the_watchdog_and_exits_3():
    p = _run(["--dry-run", "--simulate-hung-capture", "--graph-timeout", "2"])
    assert p.returncode == 3, p.stderr[-2000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["hang"]["leg"] == "hipGraph replay" and res["legs"]["hipGraph replay"] == "hung"


def test_the_hang_exit_status_can_be_overridden():
    p = _run(["--dry-run", "--simulate-hung-capture", "--graph-timeout", "1"], env_extra={"DSS2_BENCH_HANG_RC": "0"})
    assert p.returncode == 0, p.stderr[-2000:]
    assert "hang" in json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
