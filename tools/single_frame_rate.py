#!/usr/bin/env python3
"""One frame per call, back to back, frames resident in HBM (the regime of MIMOWorker::update, src/dsp/worker.h:212-224):
device time per frame by events on the launch stream, and which kernel ran.  Under `rocprofv3 --kernel-trace --stats`
the trace splits that time into the kernel's own duration and the gap between launches.

    python tools/single_frame_rate.py [--math exact|fast|both] [workload ...]      (default: both modes, ref_default c2 headline;
                                                                                    "c2@80" = c2's arrays on an 80 x 80 grid)
"""
import importlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
pkg = importlib.import_module("beamforming-lk_amd")
S = pkg.synthetic
N = 400
args = sys.argv[1:]
modes = ["exact", "fast"]
if args[:1] == ["--math"]:
    modes = ["exact", "fast"] if args[1] == "both" else [args[1]]
    args = args[2:]
for name, mode in [(n, m) for n in (args or ["ref_default", "c2", "headline"]) for m in modes]:
    base, _, res = name.partition("@")  # "c2@80": c2's arrays on an 80 x 80 grid
    spec = S.WORKLOADS[base]
    if res:
        import dataclasses

        spec = dataclasses.replace(spec, name=f"{spec.name.split(':')[0]}@{res}: {spec.n_mics} mics x {res}x{res} x 256", res=int(res))
    xyz = S.geometry(spec)
    off, frac = S.delay_table(spec, xyz)
    frames = S.make_frames(xyz, 4, seed=1)
    d_frames = torch.from_numpy(frames).cuda()
    d_p = torch.zeros((2, spec.n_pixels), dtype=torch.float32, device="cuda")
    stream = torch.cuda.Stream()
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    math = pkg.MATH_F32_EXACT if mode == "exact" else pkg.MATH_F32_FAST
    with pkg.Engine(n_pixels=spec.n_pixels, n_streams=spec.n_mics, max_batch=1, grid_columns=spec.res, math=math) as eng:
        eng.set_delay_table(off, frac)
        eng.set_active_mics(None)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            for k in range(20):
                eng.process_device(d_frames[k % 4].data_ptr(), 1, d_p[1].data_ptr(), stream.cuda_stream)
            ev[0].record(stream)
            for k in range(N):
                eng.process_device(d_frames[k % 4].data_ptr(), 1, d_p[1].data_ptr(), stream.cuda_stream)
            ev[1].record(stream)
            torch.cuda.synchronize()
        st = eng.stats()
    us = ev[0].elapsed_time(ev[1]) / N * 1e3
    flops = int(st.alg_flops_frame)
    print(f"{spec.name} [{mode}]: {pkg.binding.KERNEL_NAMES[st.kernel_variant]} {us:.2f} us per frame on the device "
          f"({flops / (us * 1e-6) / 1e12:.1f} TFLOP/s algorithmic = {flops / (us * 1e-6) / 1e12 / 157.3:.3f} of the fp32 peak)")
