#!/usr/bin/env python3
"""Throughput of the host-buffer entry point orbx_extract_batch (frames start and results end in host memory,
so every step pays the PCIe transfers): the PCIe-inclusive companion of bench.py's device-resident figure.

    python tools/bench_host_path.py [--batch 256] [--steps 8]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=8)
    a = ap.parse_args()
    import my_orb_slam2_amd as m
    frames = m.synth_batch(0, a.batch, 752, 480)
    ex = m.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=a.batch)
    ex.extract_batch(frames)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = ex.extract_batch(frames)
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "ORB frames/s through orbx_extract_batch (host buffers in and out, PCIe included)",
                      "value": round(a.batch * a.steps / dt, 1), "unit": "frames/s", "batch": a.batch, "steps": a.steps,
                      "ms_per_step": round(dt / a.steps * 1e3, 3),
                      "mean_keypoints": round(sum(len(k) for k, _ in out) / len(out), 1)}))
    # one frame at a time, as Tracking calls ORBextractor::operator() (src/Frame.cc:285-291)
    one = m.ORBextractor(1000, 1.2, 8, 20, 7)
    one(frames[0])
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        one(frames[i % a.batch])
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "latency of one orbx_extract call (752x480, host image in, host keypoints out)",
                      "value": round(dt / n * 1e3, 3), "unit": "ms", "calls": n}))


if __name__ == "__main__":
    main()
