"""`python bench.py --gpus N` starts its N ranks itself: a fresh `python -m torch.distributed.run` child
spawned BEFORE torch / HIP are imported in the parent; the ranks assert WORLD_SIZE == --gpus.  The reference's counterpart is the
world_size=1 DDP stub of train_tile.py:227-238.

CPU: both children must reach bench.py's own "needs a GPU" exit (the launcher worked, nothing fell back to a CPU path).
GPU: two ranks on the one card over gloo (RCCL refuses two ranks per device) must print ONE JSON line with n_gpus == 2 / dp2."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def test_self_launch_happens_before_torch_is_imported():
    src = open(BENCH).read()
    assert src.index("_self_launch()") < src.index("import torch  #"), "the launcher must run before torch is imported"
    assert "os.exec" not in src and "execv" not in src          # never replace a process image (GPU boxes refuse it)


def test_gpus_2_spawns_two_ranks_that_refuse_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=_clean_env(), cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "bench.py needs a GPU" in out, out[-2000:]
    assert "local_rank: 1" in out or "local_rank: 0" in out       # torchrun's failure report names a rank of the 2-rank job
    assert '"n_gpus"' not in out                                   # and no number was printed


def test_world_size_must_match_gpus_flag():
    env = _clean_env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--backend", "gloo"], capture_output=True, text=True, timeout=600,
                       env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "--gpus 4 but WORLD_SIZE=2" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_gpus_2_over_gloo_on_one_card_prints_dp2(dev):
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--no-launch-timing"], capture_output=True, text=True, timeout=900, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["steps"] == 3 and out["warmup"] == 1
    assert out["value"] > 0 and out["cpu_baseline"] is None
