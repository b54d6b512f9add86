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


def load_bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", REPO / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_batch_grows_with_the_ranks():
    """Default frames per step = 128 x ranks (c5: 1024 x ranks): a rank's launch -- B frames x 1/N of the grid --
    keeps its one-GPU size at every N, so launch and collective latency are amortised alike; --batch pins it."""
    bench = load_bench()
    assert [bench.default_batch(n, False) for n in (1, 2, 4, 8)] == [128, 256, 512, 1024]
    assert [bench.default_batch(n, True) for n in (1, 8)] == [1024, 8192]
    for n in (1, 2, 4, 8):  # per-rank work (frames x rows) is the same at every N
        assert bench.default_batch(n, False) * (128 // n) == 128 * 128
    assert bench.parse_args(["--gpus", "8"]).batch == 0 and bench.parse_args(["--batch", "64"]).batch == 64


def test_no_child_ranks_under_a_profiler():
    """Under rocprofv3 (or the repo's profiling scripts, which set AWPU_UNDER_PROFILER=1) the GPU is initialised before
    main() runs; starting torch.distributed.run from there is the exec this pool forbids.  --gpus N must refuse."""
    bench = load_bench()
    assert bench.under_profiler({}) == ""
    assert bench.under_profiler({"AWPU_UNDER_PROFILER": "1"})
    assert bench.under_profiler({"AWPU_NO_BUILD": "1"}) == ""  # only stops rebuilds (tools/gpu_ab_libs.sh sets it with no profiler about)
    assert bench.under_profiler({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so"})
    assert bench.under_profiler({"ROCPROF_OUTPUT_PATH": "/tmp/x"}) and bench.under_profiler({"ROCP_TOOL_LIB": "x"})
    import os

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["AWPU_UNDER_PROFILER"] = "1"
    proc = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--selftest-launcher", "ok"], env=env,
                          capture_output=True, text=True, timeout=120)
    assert proc.returncode == 2 and "refusing to start child ranks" in proc.stderr
    assert not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for script in ("pmc.sh", "pmc_hbm.sh", "gpu_profile.sh"):
        text = (REPO / "tools" / script).read_text()
        assert "--gpus" in text, f"tools/{script} must reject --gpus"
        assert "export AWPU_UNDER_PROFILER=1" in text, f"tools/{script} must mark its processes as profiled"


def test_watchdog_exits_non_zero():
    """A rendezvous / first collective that never returns ends the rank with status 4 instead of hanging."""
    code = ("import sys, time; sys.path.insert(0, %r); import importlib.util as u; "
            "s = u.spec_from_file_location('b', %r); m = u.module_from_spec(s); s.loader.exec_module(m)\n"
            "with m.Watchdog(0.5, 'test rendezvous'):\n    time.sleep(30)\n") % (str(REPO), str(REPO / "bench.py"))
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert proc.returncode == 4 and "did not finish" in proc.stderr
