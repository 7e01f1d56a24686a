"""`python bench.py --gpus N` launches itself (round-4 review item 6): the parent starts N ranks before anything touches a GPU,
gives them the environment torch.distributed.run would, relays rank 0's JSON line and fails when a rank does.  The ranks here
only prove that they exist (CSN_BENCH_LAUNCH_CHECK: gloo on the CPU) — the launch logic needs no GPU."""
import json
import os
import re
import signal
import subprocess
import sys
import time

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


def _gone(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return True
    except PermissionError:
        return False
    try:                                                              # a zombie of a dead parent's child still answers kill 0
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[0] == "Z"
    except OSError:
        return True


def test_a_rank_that_dies_before_a_collective_does_not_hang_the_run():
    """SURVEY §8(e) / round-5 review: a rank dies BEFORE a collective while the others sit in it (the check's other ranks refuse to
    return, like ranks inside an RCCL collective).  The parent notices the first non-zero exit, stops the others and returns
    non-zero within seconds — not after a collective timeout."""
    t0 = time.monotonic()
    res = _run({"CSN_BENCH_LAUNCH_CHECK": "die_before_collective"}, "--gpus", "3")
    took = time.monotonic() - t0
    assert res.returncode != 0 and "ranks failed" in res.stderr and "(2, 3)" in res.stderr, res.stderr[-2000:]
    assert took < 60, took
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]         # no result line from a failed run
    pids = [int(m.group(1)) for m in re.finditer(r"bench-rank-pid \d+ (\d+)", res.stderr)]
    assert len(pids) == 3
    time.sleep(0.5)
    assert all(_gone(p) for p in pids), [p for p in pids if not _gone(p)]       # nothing keeps holding a GPU


def test_stopping_the_parent_stops_the_ranks():
    """`timeout` (or the driver) sends the parent SIGTERM: every rank goes with it."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["CSN_BENCH_LAUNCH_CHECK"] = "sleep"
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE)
    seen, pids, t0 = "", [], time.monotonic()
    os.set_blocking(proc.stderr.fileno(), False)
    while seen.count("bench-rank-ready") < 2 and time.monotonic() - t0 < 240:
        try:
            seen += os.read(proc.stderr.fileno(), 65536).decode(errors="replace")
        except BlockingIOError:
            time.sleep(0.1)
    assert seen.count("bench-rank-ready") == 2, seen[-2000:]
    pids = [int(m.group(1)) for m in re.finditer(r"bench-rank-pid \d+ (\d+)", seen)]
    t1 = time.monotonic()
    proc.send_signal(signal.SIGTERM)
    proc.wait(timeout=60)
    assert proc.returncode == 128 + signal.SIGTERM and time.monotonic() - t1 < 30
    time.sleep(0.5)
    assert len(pids) == 2 and all(_gone(p) for p in pids)


def test_launch_deadline():
    """CSN_BENCH_LAUNCH_TIMEOUT_S: a run in which every rank stays alive but nothing comes out is cut by the launcher itself."""
    t0 = time.monotonic()
    res = _run({"CSN_BENCH_LAUNCH_CHECK": "sleep", "CSN_BENCH_LAUNCH_TIMEOUT_S": "25"}, "--gpus", "2")
    assert res.returncode != 0 and "CSN_BENCH_LAUNCH_TIMEOUT_S" in res.stderr and time.monotonic() - t0 < 90


def test_a_taken_rendezvous_port_gets_one_more_attempt():
    """The launcher picks its port by bind-and-close; if somebody takes it before rank 0 binds, the ranks fail at rendezvous within
    seconds: one more attempt with another port (here the first port is one this test holds)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        sk.listen(1)
        res = _run({"CSN_BENCH_LAUNCH_CHECK": "1", "CSN_BENCH_FIRST_PORT": str(sk.getsockname()[1])}, "--gpus", "2")
    assert res.returncode == 0, res.stderr[-2000:]
    assert "one more attempt" in res.stderr
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_ranks_seen"] == 2
