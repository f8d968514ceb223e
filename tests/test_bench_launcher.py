"""bench.py called plainly with --gpus N starts its own rank processes before anything touches a GPU (VERDICT r1, next
#3).  The launcher itself is CPU code: a hidden --launch-check makes every rank report its environment and exit."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_plain_call_with_gpus_n_starts_n_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--launch-check"], env=_env(), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=60)
    assert out.returncode == 0, out.stderr.decode()
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1  # only rank 0 owns stdout
    e = json.loads(lines[0])
    assert e["RANK"] == "0" and e["LOCAL_RANK"] == "0" and e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1"
    assert int(e["MASTER_PORT"]) > 0


def test_a_failing_rank_stops_the_others_and_sets_the_exit_status():
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--launch-check", "--launch-check-fail", "1"],
                         env=dict(_env(), BENCH_LAUNCH_GRACE="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert out.returncode == 3
    assert time.monotonic() - t0 < 30  # the sleeping peers were stopped, not waited for
    assert b"rank 1 exited with status 3" in out.stderr


def test_under_a_launcher_the_world_size_must_match():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=dict(_env(), WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert out.returncode != 0 and b"WORLD_SIZE is 4" in out.stderr
