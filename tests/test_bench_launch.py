"""`python bench.py --gpus N` launches itself (round-4 review item 6): the parent starts N ranks before anything touches a GPU,
gives them the environment torch.distributed.run would, relays rank 0's JSON line and fails when a rank does.  The ranks here
only prove that they exist (CSN_BENCH_LAUNCH_CHECK: gloo on the CPU) — the launch logic needs no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)


def test_gpus_2_spawns_two_ranks_and_relays_rank_0():
    res = _run({"CSN_BENCH_LAUNCH_CHECK": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                            # ONE JSON line, rank 0's
    out = json.loads(lines[0])
    assert out["n_ranks_seen"] == 2 and out["rank_sum"] == 3.0 and out["local_rank"] == 0
    assert out["argv"] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]      # every rank gets the parent's arguments


def test_a_rank_that_dies_fails_the_run():
    res = _run({"CSN_BENCH_LAUNCH_CHECK": "fail"}, "--gpus", "3")
    assert res.returncode != 0 and "ranks failed" in res.stderr


def test_under_a_launcher_nothing_is_spawned():
    """WORLD_SIZE in the environment = torch.distributed.run (or the parent above) already made the ranks: a rank whose world
    is not --gpus still refuses to run."""
    res = _run({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "CSN_BENCH_LAUNCH_CHECK": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29617"},
               "--gpus", "1")
    assert res.returncode == 0 and json.loads(res.stdout.strip().splitlines()[-1])["n_ranks_seen"] == 1
