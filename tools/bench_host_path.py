#!/usr/bin/env python3
"""Throughput of the host-buffer entry point orbx_extract_batch (frames start and results end in host memory,
so every step pays the PCIe transfers): the PCIe-inclusive companion of bench.py's device-resident figure.

    python tools/bench_host_path.py [--batch 256] [--steps 8]

Also measures the arrangement a throughput caller would use instead: frames and results in pinned host memory,
three slots (each an extractor context and a stream of its own), copy in -> orbx_extract_batch_device -> copy out
queued on the slot's stream, so the transfers of one batch overlap the kernels of another.
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
    del ex
    # pinned memory + the device entry point, three batches in flight
    import torch
    B, slots = a.batch, 3
    exs = [m.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B) for _ in range(slots)]
    cap = exs[0].capacity
    streams = [torch.cuda.Stream() for _ in range(slots)]
    h_in = [torch.from_numpy(frames).clone().pin_memory() for _ in range(slots)]
    d_in = [torch.empty((B, 480, 752), dtype=torch.uint8, device="cuda") for _ in range(slots)]
    d_out = [(torch.empty((B, cap, 7), dtype=torch.float32, device="cuda"),
              torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda"),
              torch.zeros(B, dtype=torch.int32, device="cuda")) for _ in range(slots)]
    h_out = [tuple(torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in d_out[0]) for _ in range(slots)]

    def submit(i):
        k = i % slots
        st = streams[k]
        st.synchronize()                                 # the slot's previous batch is back in host memory
        with torch.cuda.stream(st):
            d_in[k].copy_(h_in[k], non_blocking=True)
            exs[k].extract_batch_device(d_in[k], *d_out[k], stream=st.cuda_stream)
            for h, d in zip(h_out[k], d_out[k]):
                h.copy_(d, non_blocking=True)
    for i in range(slots):
        submit(i)
    torch.cuda.synchronize()
    steps = max(a.steps, 4) * 4
    t0 = time.perf_counter()
    for i in range(steps):
        submit(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "ORB frames/s, pinned host memory in and out, 3 batches in flight (PCIe included)",
                      "value": round(B * steps / dt, 1), "unit": "frames/s", "batch": B, "steps": steps,
                      "ms_per_step": round(dt / steps * 1e3, 3),
                      "bytes_per_step": {"in": int(h_in[0].numel()), "out": int(sum(t.numel() * t.element_size() for t in h_out[0]))},
                      "mean_keypoints": round(float(h_out[0][2].float().mean()), 1)}))
    del exs
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
