"""bench.py --gpus N starts its N ranks by itself (child processes through torch.distributed.run, rendezvous on
127.0.0.1) when no launcher is around it - the form the round driver uses.  CPU test: the launch-only hook runs the
launcher, the rendezvous (gloo) and one collective, and prints a line with n_gpus = N; no GPU work."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [2, 3])
def test_bench_gpus_n_spawns_n_ranks(n):
    env = dict(os.environ, VFT_BENCH_LAUNCH_ONLY="1", VFT_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout.decode()
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["world"] == n
    assert line["rank_sum"] == n * (n + 1) / 2      # every rank took part in the collective


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, VFT_BENCH_LAUNCH_ONLY="1", WORLD_SIZE="1", RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=120)
    assert res.returncode != 0 and b"WORLD_SIZE" in res.stderr


def test_a_failing_rank_fails_the_launcher():
    env = dict(os.environ, VFT_BENCH_LAUNCH_ONLY="1", VFT_BENCH_BACKEND="no-such-backend")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert res.returncode != 0


def test_single_gpu_launcher_merges_its_children_and_survives_a_dying_leg():
    """At N = 1 bench.py is a launcher that never touches the GPU: the step measurement and every end-to-end leg are child processes,
    their records are merged into one line, and a leg whose process dies (here: killed by the test hook) costs its own record only.
    The line is printed after the step measurement and again after every leg, each a superset of the one before (whatever ends the
    launcher early, what was measured by then is on stdout): the last line is the result."""
    env = dict(os.environ, VFT_BENCH_FAKE_CHILD="e2e_c2_threads")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-e2e-c5-one-thread"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [json.loads(ln) for ln in res.stdout.decode().splitlines() if ln.startswith("{")]
    legs = ["e2e", "e2e_c4", "e2e_c2", "e2e_c2_threads", "e2e_c5_threads", "e2e_c4s_threads", "e2e_c4_full_threads"]
    assert len(lines) == 1 + len(legs)
    assert lines[0]["legs_pending"] == legs and not [k for k in lines[0] if k.startswith("e2e")]
    for a, b in zip(lines, lines[1:]):   # every line a superset of the one before
        assert all(k in b for k in a) and all(b[k] == a[k] for k in a if k != "legs_pending")
    line = lines[-1]
    assert line["metric"] == "profile-ops/sec" and line["legs_pending"] == []
    assert [k for k in line if k.startswith("e2e")] == legs
    assert "error" in line["e2e_c2_threads"] and line["e2e_c5_threads"]["wall_s"] == 1.0


def test_a_leg_that_would_pass_the_time_budget_is_skipped_not_started():
    """--time-budget: a leg is skipped when the time used so far plus what the leg took last round would pass the budget (the driver's clock
    must not run out inside a leg); the line says so and the later, shorter legs still run."""
    env = dict(os.environ, VFT_BENCH_FAKE_CHILD="none")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--time-budget", "200"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    line = [json.loads(ln) for ln in res.stdout.decode().splitlines() if ln.startswith("{")][-1]
    assert "skipped" in line["e2e_c4_full_threads"] and "skipped" in line["e2e_c5"]
    assert line["e2e"]["wall_s"] == 1.0 and line["e2e_c2_threads"]["wall_s"] == 1.0
