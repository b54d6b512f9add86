#!/usr/bin/env python3
"""kbench.py -- time kernel variants of the fast sweep on one GPU (tuning aid, not the bench).

    python tools/kbench.py [--workload headline] [--batch 64] variants...   e.g.  1,8 1,4 2,4

Each variant "fpi,ppw" runs bench.py in its own process with AWPU_FAST_VARIANT set and prints the
kernel time per launch, frames/s and the parity error bench.py measured.
"""
import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="headline")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("variants", nargs="*", default=["1,8"])
args = ap.parse_args()

for v in args.variants:
    env = dict(os.environ, AWPU_FAST_VARIANT=v)
    cmd = [sys.executable, str(REPO / "bench.py"), "--steps", str(args.steps), "--warmup", "2", "--batch",
           str(args.batch), "--workload", args.workload, "--cpu-seconds", "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(v, "FAILED", out.stderr[-400:])
        continue
    j = json.loads(line[-1])
    print(f"variant {v:>5}  batch {args.batch:3d}  kernel {j['roofline']['kernel_ms']:8.3f} ms  "
          f"{j['value']:9.1f} frames/s  us/frame {1e3 * j['roofline']['kernel_ms'] / args.batch:7.2f}  "
          f"valu {j['valu']['frac'] * 100:5.1f}%  parity {j['parity_max_rel_err']:.2e}", flush=True)
