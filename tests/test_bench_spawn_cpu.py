"""`python bench.py --gpus N` without a torchrun environment starts its N ranks itself (bench.py:spawn_ranks) BEFORE anything
touches a GPU.  No 8-GPU node is available to the builder, so the plumbing is exercised here on the CPU with the rank body
replaced by a gloo stub (SFM_BENCH_DRYRUN=1): the composed `torch.distributed.run` command works, the ranks rendezvous on
127.0.0.1, the per-step collective runs, ONLY rank 0 prints the JSON line, and a failing rank's exit code reaches the caller.
Reference mechanism being replaced: the stock Chainer updaters chosen by YAML (config_utils.py:122-133,156-161)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, extra_env=None, launcher=False):
    env = dict(os.environ, SFM_BENCH_DRYRUN="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env.update(extra_env or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"]
    if launcher:   # the driver's own form
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", "29617"] + cmd[1:]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)


def json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_bench_starts_its_ranks_itself_and_rank0_alone_reports():
    r = run(2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout            # one line, from rank 0
    assert lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 3 and lines[0]["sum_of_ranks"] == 3.0   # 1 + 2: the collective ran over both ranks


def test_bench_starts_eight_ranks():
    """BASELINE cfg4's size (round-5 verdict item 5): `python bench.py --gpus 8` composes the launcher command for eight ranks, they
    rendezvous on 127.0.0.1, the per-step collective sums over all eight (1 + 2 + ... + 8), rank 0 alone reports."""
    r = run(8)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 8 and lines[0]["sum_of_ranks"] == 36.0


def test_bench_under_the_drivers_launcher():
    r = run(2, launcher=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_a_failing_rank_fails_the_run():
    r = run(2, {"SFM_BENCH_DRYRUN_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not json_lines(r.stdout) or True     # (rank 0 may or may not have printed before the launcher tore it down)


def test_world_size_mismatch_is_refused():
    r = run(2, {"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)


def test_a_rank_without_a_device_says_so_and_fails():
    """"if the box exposes fewer devices than ranks, report the devices found".  On this host there are none: the real
    rank body (no dry run) must leave with one clear line and a non-zero code, before any HIP call, instead of a bare runtime error."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "SFM_BENCH_DRYRUN"):
        env.pop(k, None)
    import torch
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip("this host has a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "exposes 0 GPU(s)" in (r.stderr + r.stdout), (r.stdout[-500:], r.stderr[-500:])
    assert not json_lines(r.stdout)
