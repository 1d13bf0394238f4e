"""The N > 1 path of bench.py with real kernels, on a box with ONE GPU: `python bench.py --gpus 2` under
SFM_BENCH_REHEARSE_ONE_GPU=1 starts its two ranks itself (before anything touches the GPU), both run their shard on cuda:0, the
per-step collective goes through gloo (RCCL refuses two ranks on one device), rank 0 prints the line.  Not a measurement: what is
checked is that the path runs and that the all-reduced scalars are the global loss."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu(dev):
    env = dict(os.environ, SFM_BENCH_REHEARSE_ONE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-secondary",
                        "--no-cpu-baseline", "--min-time", "0.01"], env=env, capture_output=True, text=True, timeout=580, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]                # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and "rehearsal" in d
    assert d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"]
    assert "per_step_collective_variant" in d and "allreduce_160MB" in d and "once per block" in d["config"]["parallelism"]
    total, pixel, smooth, expl, ssim = d["loss5"]
    # every rank normalises by the GLOBAL batch, so the summed scalars are a loss of ordinary size (not twice / half of one)
    assert 1.0 < total < 8.0 and abs(total - (0.85 * pixel + 0.15 * ssim + smooth + expl)) < 1e-3 * total
