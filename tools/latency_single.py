#!/usr/bin/env python3
"""Latency of the drop-in call itself: one 752x480 frame per call (what the tracking thread does), host buffer
in, host buffers out. Prints the wall time per call, the GPU time of the four stages and the same for the
device-resident single-frame call.

    python tools/latency_single.py [--calls 2000]
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
    ap.add_argument("--calls", type=int, default=2000)
    ap.add_argument("--width", type=int, default=752)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    a = ap.parse_args()
    import numpy as np
    import torch
    import my_orb_slam2_amd as m

    W, H = a.width, a.height
    ex = m.ORBextractor(a.nfeatures, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1)
    frames = [m.synth_frame(i, W, H) for i in range(8)]
    for i in range(50):
        ex(frames[i % 8])
    t0 = time.perf_counter()
    n = 0
    for i in range(a.calls):
        k, d = ex(frames[i % 8])
        n += len(k)
    host_ms = (time.perf_counter() - t0) / a.calls * 1e3
    # the C entry point alone, outputs into arrays made once (what a C++ caller sees; the Python binding above
    # also allocates and copies its results)
    import ctypes as C
    from my_orb_slam2_amd._lib import lib, np_ptr
    kps = np.empty(ex.capacity, dtype=m.KEYPOINT_DTYPE)
    desc = np.empty((ex.capacity, 32), dtype=np.uint8)
    cnt = C.c_int(0)
    fn = lib().orbx_extract
    args = [(np_ptr(f), f.strides[0]) for f in frames]
    t0 = time.perf_counter()
    for i in range(a.calls):
        fp, st = args[i % 8]
        fn(ex._ctx, fp, W, H, st, np_ptr(kps), np_ptr(desc), ex.capacity, C.byref(cnt))
    host_c_ms = (time.perf_counter() - t0) / a.calls * 1e3

    cap = ex.capacity
    dimg = torch.from_numpy(np.stack(frames)).cuda()
    kp = torch.empty((1, cap, 7), dtype=torch.float32, device="cuda")
    ds = torch.empty((1, cap, 32), dtype=torch.uint8, device="cuda")
    cn = torch.zeros(1, dtype=torch.int32, device="cuda")
    for i in range(50):
        ex.extract_batch_device(dimg[i % 8:i % 8 + 1], kp, ds, cn)
    ex.synchronize()
    t0 = time.perf_counter()
    for i in range(a.calls):
        ex.extract_batch_device(dimg[i % 8:i % 8 + 1], kp, ds, cn)
        ex.synchronize()
    dev_ms = (time.perf_counter() - t0) / a.calls * 1e3
    t0 = time.perf_counter()
    for i in range(a.calls):
        ex.extract_batch_device(dimg[i % 8:i % 8 + 1], kp, ds, cn)
    ex.synchronize()
    dev_pipe_ms = (time.perf_counter() - t0) / a.calls * 1e3
    ex.set_profiling(True)
    st = {}
    for i in range(200):
        ex.extract_batch_device(dimg[i % 8:i % 8 + 1], kp, ds, cn)
        ex.synchronize()
        for name, ms in ex.stage_ms().items():
            st[name] = st.get(name, 0.0) + ms / 200
    print(json.dumps({"workload": "%dx%d, %d keypoints, one frame per call" % (W, H, a.nfeatures), "calls": a.calls,
                      "mean_keypoints": n / a.calls,
                      "host_call_ms": round(host_ms, 4), "host_call_c_abi_ms": round(host_c_ms, 4), "device_call_sync_ms": round(dev_ms, 4),
                      "device_call_back_to_back_ms": round(dev_pipe_ms, 4),
                      "stage_ms": {k: round(v, 4) for k, v in st.items()}}))


if __name__ == "__main__":
    main()
