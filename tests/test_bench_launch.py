"""`python bench.py --gpus N` must produce N ranks by itself: without WORLD_SIZE in the environment the
process becomes a supervisor that starts N fresh children before anything touches the GPU.  The same launch path runs
here on CPU: SFA_BENCH_LAUNCH_TEST=1 makes the ranks rendezvous over gloo, count themselves and stop before the GPU part."""
import json
import os
import subprocess
import sys

from tests.util import ROOT


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_bench_spawns_its_own_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, timeout=300, cwd=ROOT, env=_env(SFA_BENCH_LAUNCH_TEST="1"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]  # (gloo logs its connections on stdout)
    assert len(lines) == 1, lines  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d == {"launch_test": True, "n_gpus": 2, "world": 2}


def test_strong_scaling_splits_one_job_over_the_ranks():
    """--total-reads: contiguous, disjoint ranges that cover the job, whatever the remainder (two gloo ranks, no GPU)."""
    for total, want in ((1_000_000, [[0, 500_000], [500_000, 1_000_000]]), (7, [[0, 3], [3, 7]])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--total-reads", str(total),
                            "--workload", "r10_dna_1mb_q250"], capture_output=True, timeout=300, cwd=ROOT, env=_env(SFA_BENCH_LAUNCH_TEST="1"))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        d = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
        assert d["scaling"] == "strong" and d["read_ranges"] == want and d["world"] == 2


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    # a launcher that produced ONE rank while --gpus says 2: an error, never a 1-GPU number under a 2-GPU label
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, timeout=300,
                       cwd=ROOT, env=_env(SFA_BENCH_LAUNCH_TEST="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0
    assert b"--gpus 2 but 1 rank" in r.stderr
    assert not r.stdout.strip()


def test_a_failing_rank_fails_the_launch():
    from sigfish_amd import launch
    code = "import os,sys,time; sys.exit(3) if os.environ['RANK']=='1' else time.sleep(30)"
    rc = launch.spawn_ranks(2, ["-c", code], timeout=60)
    assert rc == 3
