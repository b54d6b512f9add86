"""bench.py --gpus N starts its own ranks when it was not started by torch.distributed.run: the launcher
path on CPU (gloo), with the ranks running a rendezvous + one collective instead of the GPU bench."""
import json
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def run_bench(*argv):
    env = {k: v for k, v in __import__("os").environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, str(REPO / "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=300)


def test_self_launch_two_ranks_one_line():
    proc = run_bench("--gpus", "2", "--selftest-launcher", "ok")
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    rec = json.loads(lines[0])
    assert rec == {"launcher_selftest": True, "world": 2, "sum": 3.0}


def test_self_launch_reports_a_failing_rank():
    proc = run_bench("--gpus", "2", "--selftest-launcher", "fail")
    assert proc.returncode != 0
    assert not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_without_gpu():
    """No GPU, no number: the bench has no CPU path to fall back to."""
    import torch

    if torch.cuda.is_available():
        return
    proc = run_bench("--steps", "1", "--warmup", "0", "--cpu-seconds", "0")
    assert proc.returncode != 0 and "needs a GPU" in (proc.stderr + proc.stdout)
