#!/usr/bin/env python3
"""replay_sender.py -- plays the FPGA (or `udpreplay` of a capture) for the live path: UDP datagrams in the
reference's wire format (src/fpga/receiver.h:24-30: u16 frequency, u8 n_arrays, u8 version, u32 counter,
i32 stream[256], packed, 1032 bytes), one per sample time, carrying the reference's synthetic signal
(src/fpga/pipeline.cpp:105-135: a 9 kHz plane wave, here from an off-axis direction) as 24-bit samples in
wire order (every other group of 8 columns mirrored, pipeline.cpp:277-287).

    python tools/replay_sender.py --port 21844 [--address 127.0.0.1] [--arrays 1] [--rate 1.0] [--seconds 5]

--rate is relative to real time (48828 datagrams/s); 0 = as fast as the socket takes them.  The receiving
side is awpu_host::PipelineHip (beamforming-lk_amd/host/pipeline_hip.h) or awpu_hip_ingest_block directly.
"""
import argparse
import importlib
import socket
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

SAMPLE_RATE = 48828.0
MSG = np.dtype([("frequency", "<u2"), ("n_arrays", "u1"), ("version", "u1"), ("counter", "<u4"), ("stream", "<i4", (256,))])


def wire_order(n_sensors: int) -> np.ndarray:
    """wire slot of sensor s (pipeline.cpp:277-287: `inverted` toggles at every multiple of 8, starting inverted)"""
    s = np.arange(n_sensors)
    return np.where((s // 8) % 2 == 0, 8 * (1 + s // 8) - 1 - s % 8, s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--address", default="127.0.0.1")
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--arrays", type=int, default=1)
    ap.add_argument("--rate", type=float, default=1.0)
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--theta-deg", type=float, default=20.0)
    ap.add_argument("--phi-deg", type=float, default=35.0)
    args = ap.parse_args()

    pkg = importlib.import_module("beamforming-lk_amd")
    n = 64 * args.arrays
    xyz = pkg.create_tiled_antenna(args.arrays, 1)
    tau = pkg.steering_delays(xyz, np.deg2rad(args.theta_deg), np.deg2rad(args.phi_deg)).astype(np.float64)
    slot = wire_order(n)
    tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    block = np.zeros(256, MSG)
    block["frequency"], block["n_arrays"], block["version"] = 48828, args.arrays, 2
    t0 = time.perf_counter()
    p = 0
    while time.perf_counter() - t0 < args.seconds:
        t = (p + np.arange(256))[:, None] + tau[None, :]                      # [sample][sensor]
        v = 1e-2 * np.sin(2.0 * np.pi * 9e3 * t / SAMPLE_RATE)
        block["stream"][:, slot] = np.rint(v * 8388608.0).astype(np.int32)     # sensor s travels in slot[s]
        block["counter"] = p + np.arange(256)
        raw = block.tobytes()
        for i in range(256):
            tx.sendto(raw[i * 1032:(i + 1) * 1032], (args.address, args.port))
        p += 256
        if args.rate > 0:
            ahead = p / (SAMPLE_RATE * args.rate) - (time.perf_counter() - t0)
            if ahead > 0:
                time.sleep(ahead)
    print(f"sent {p} datagrams ({p // 256} blocks) in {time.perf_counter() - t0:.2f} s")


if __name__ == "__main__":
    main()
