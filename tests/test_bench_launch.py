"""`python bench.py --gpus N` as the driver invokes it for N = 1: a plain interpreter, no launcher around it.  For N > 1 the script
itself must start the ranks as child processes (before it touches the GPU), relay rank 0's line and fail loudly when a rank fails."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_plain_invocation_starts_the_ranks_itself():
    r = _run({"MIS_BENCH_LAUNCH_CHECK": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout            # ONE line on stdout, whatever the launcher and the ranks print
    res = json.loads(lines[0])
    assert res["launch_check"] is True and res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1
    assert res["master_addr"] == "127.0.0.1"


def test_a_failing_rank_fails_the_run():
    r = _run({"MIS_BENCH_LAUNCH_CHECK": "1", "MIS_BENCH_LAUNCH_CHECK_FAIL_RANK": "1"}, "--gpus", "2")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_world_size_mismatch_is_refused():
    env = {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env={**os.environ, **env}, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.gpu
def test_two_rank_rehearsal_from_a_plain_invocation_on_one_gpu():
    """The real N > 1 branch of bench.py (sharded job, exchange, roofline leg on rank 0) started by `python bench.py --gpus 2`;
    both ranks on the one GPU of the box, gloo instead of RCCL (MIS_BENCH_REHEARSAL): a functional check, not a measurement."""
    r = _run({"MIS_BENCH_REHEARSAL": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "config3", "--no-single-base",
             timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["frames"] == 16 and res["value"] > 0
    assert res["roofline"]["parts"]["warp"]["frac"] > 0
    # the C++ host's sharded job on two child ranks of host/stitch_bench (exchanges staged through shared memory in a one-GPU rehearsal)
    cpp = res["cpp_host"]
    assert "error" not in cpp, cpp
    assert "ShardedJob, 2 ranks" in cpp["host"] and cpp["frames"] == 16 and cpp["kept"] == 16 and cpp["pano_size"] == res["config"]["pano_size"]
