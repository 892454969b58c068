#!/usr/bin/env python3
"""Throughput of the ORB front end on MI355X.

Workload (BASELINE.json configs[1]): a synthetic batch of 752x480 uint8 frames, ORB extract+describe
with nFeatures 1000, 8 levels, scale 1.2, FAST 20/7.  One "step" = one batch of --batch frames that
already sits in HBM through the whole hot path (pyramid, FAST cells, quadtree, orientation + BRIEF);
outputs stay on the device.  N > 1: one process per GPU, every rank extracts its own frames, no
data-path collective ("scaling": "weak").

    python bench.py --gpus 1 --steps 256 --warmup 16
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `roofline` is for the kernel that takes most of a step, timed with HIP
events on the library's launch stream; `cpu_baseline` times the CPU oracle (a port of the reference's
algorithm; the reference itself needs OpenCV 4 and cannot be built here) on the host cores.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (spec)
W, H, NFEAT, NLEV, SCALE, INI, MINTH = 752, 480, 1000, 8, 1.2, 20, 7
# Algorithmic bytes per 752x480 frame: every pyramid level read once (1,117,367) + every derived level
# written once (756,407) + outputs 1000 x (28 + 32).
BYTES_PER_FRAME = 1117367 + 756407 + 60000
# The same split by stage: what each kernel has to move at the very least.
LEVEL_PX = 1117367
DERIVED_PX = 756407
STAGE_BYTES = {
    "pyramid": (LEVEL_PX - 210 * 134) + DERIVED_PX,     # read levels 0..6, write levels 1..7
    "fast_cells": LEVEL_PX,                             # read every level once (+ a few KB of candidates)
    "octree": 4 * 10000 + 4 * 1000,                     # ~10k packed candidates in, ~1k selected out
    "orient_desc": 1000 * (43 * 43 + 60),               # one 43x43 patch per keypoint + its outputs
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=256, help="frames per step and GPU")
    ap.add_argument("--unique", type=int, default=1024, help="distinct synthetic frames per GPU (cycled)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-hamming", action="store_true")
    ap.add_argument("--contexts", type=int, default=3,
                    help="extractor contexts (HIP streams) a GPU alternates between; the tail of one step's "
                         "kernels overlaps the head of the next step's")
    return ap.parse_args()


def cpu_baseline(seconds, frames):
    """Frames/s of the CPU oracle, one frame per thread (the reference's unit of parallelism is one
    extractor call per thread, src/Frame.cc:97-110)."""
    from tests import oracle as orc
    cores = min(16, os.cpu_count() or 1)
    done = [0] * cores
    stop = time.perf_counter() + seconds

    def work(t):
        ex = orc.Extractor(NFEAT, SCALE, NLEV, INI, MINTH)
        i = t
        while time.perf_counter() < stop:
            ex.extract(frames[i % len(frames)], fast=True)
            done[t] += 1
            i += cores

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    all_core = sum(done) / (time.perf_counter() - t0)
    ex = orc.Extractor(NFEAT, SCALE, NLEV, INI, MINTH)
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < min(3.0, seconds):
        ex.extract(frames[n1 % len(frames)], fast=True)
        n1 += 1
    single = n1 / (time.perf_counter() - t0)
    return {"value": round(all_core, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "single_core": round(single, 2),
            "sample": "%d distinct synthetic 752x480 frames, %d extractions in %.0f s on %d threads "
                      "(oracle/liborb_oracle.so, g++ -O3)" % (len(frames), sum(done), seconds, cores)}


def hamming_bench(torch, device):
    """Secondary metric: brute-force Hamming top-2, 65536 queries x 4M database rows on one GPU."""
    import my_orb_slam2_amd as m
    nq, ndb = 65536, 4 * 1024 * 1024
    g = torch.Generator(device="cuda").manual_seed(1)
    db = torch.randint(0, 256, (ndb, 32), dtype=torch.uint8, device="cuda", generator=g)
    q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device="cuda", generator=g)
    q[: nq // 2] = db[torch.randint(0, ndb, (nq // 2,), device="cuda", generator=g)]
    idx = torch.empty(nq, dtype=torch.int64, device="cuda")
    d1 = torch.empty(nq, dtype=torch.int32, device="cuda")
    d2 = torch.empty(nq, dtype=torch.int32, device="cuda")
    hm = m.HammingMatcher(device)
    hm.knn2_device(q, db, idx, d1, d2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        hm.knn2_device(q, db, idx, d1, d2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    pairs = nq * ndb / dt
    # VALU bound: 8 v_xor + 8 v_bcnt per pair. Integer VALU issues one wave64 instruction per SIMD every
    # 4 cycles (measured: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.0), i.e. 64 lanes/clk/CU:
    # 256 CUs x 64 lanes x 2.4 GHz = 39.3e12 lane-ops/s, 16 of them per pair.
    peak = 256 * 64 * 2.4e9 / 16
    assert int((d1[: nq // 2] == 0).sum()) == nq // 2
    out = {"metric": "Hamming pairs/s (top-2 kNN, 65536 x 4194304, 1 GPU)", "value": pairs,
           "ms": dt * 1e3, "valu_peak_pairs_per_s": peak, "frac_of_valu_peak": pairs / peak}
    out["cpu_baseline"] = hamming_cpu_baseline(q[:4096].cpu().numpy(), db[: 1 << 20].cpu().numpy())
    return out


def hamming_cpu_baseline(q, db, seconds=2.0):
    """The same top-2 scan by the CPU oracle on a bounded sample: the reference's SWAR bit count
    (src/ORBmatcher.cc:1705-1711) on one core, hardware popcount on one core and on all host threads used."""
    from tests import oracle as orc
    cores = min(16, os.cpu_count() or 1)
    chunk = 16                                   # queries per call: 16 x 1 M pairs, a few tens of milliseconds

    def run(popcnt, threads):
        done = [0] * threads
        stop = time.perf_counter() + seconds

        def work(t):
            i = t
            while time.perf_counter() < stop:
                s0 = (i * chunk) % (len(q) - chunk)
                orc.knn2(q[s0:s0 + chunk], db, 256, popcnt)
                done[t] += chunk * len(db)
                i += threads
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        [t.start() for t in th]
        [t.join() for t in th]
        return sum(done) / (time.perf_counter() - t0)
    return {"unit": "Hamming pairs/s", "kind": "port", "swar_single_core": run(False, 1),
            "popcount_single_core": run(True, 1), "value": run(True, cores), "cores": cores,
            "sample": "blocks of %d queries x %d database rows for %.0f s per figure" % (chunk, len(db), seconds)}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    import my_orb_slam2_amd as m

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Native libraries write to file descriptor 1 on their own (RCCL prints a version banner when its first
    # communicator comes up): until the result line, everything that goes to fd 1 is sent to stderr instead.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the measured path)")
    torch.cuda.set_device(local)
    # ORBX_BENCH_FORCE_DIST=1 runs the collective code path with a single rank (rehearsal on a one-GPU box)
    use_dist = world > 1 or os.environ.get("ORBX_BENCH_FORCE_DIST") == "1"
    B = a.batch
    unique = max(B, (a.unique + B - 1) // B * B)

    # ---- inputs: distinct frames per rank, resident in HBM before the clock starts ----
    host = m.synth_batch(rank * unique, unique, W, H)
    d_frames = torch.from_numpy(host).cuda()
    nbatches = unique // B
    nctx = max(1, a.contexts)
    exs = [m.ORBextractor(NFEAT, SCALE, NLEV, INI, MINTH, device=local, max_width=W, max_height=H, max_batch=B)
           for _ in range(nctx)]
    ex = exs[0]
    cap = ex.capacity
    # The process group comes up after the extractor contexts: their streams are created first and get hardware
    # queues of their own (the box runs four per process); RCCL's streams, idle during the timed region, share.
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    outs = [(torch.empty((B, cap, 7), dtype=torch.float32, device="cuda"),
             torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda"),
             torch.zeros(B, dtype=torch.int32, device="cuda")) for _ in range(nctx)]
    d_cnt = outs[0][2]

    def step(i):
        b = i % nbatches
        k, d, c = outs[i % nctx]
        exs[i % nctx].extract_batch_device(d_frames[b * B:(b + 1) * B], k, d, c)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kp_mean = float(d_cnt.float().mean().item())

    # ---- per-stage kernel time, HIP events on the library's launch stream (outside the timed region) ----
    ex.set_profiling(True)
    acc = {}
    reps = 8
    for i in range(reps):
        step(nctx * i)                      # context 0 only, so stages do not overlap another step
        for k, v in ex.stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / reps
    ex.set_profiling(False)

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return
    fps = world * B * a.steps / dt
    dom = max(acc, key=acc.get)
    dom_bytes = STAGE_BYTES[dom] * B
    achieved = dom_bytes / (acc[dom] * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(dom)
    # what actually bounds the kernel (profiles/valu.json, from the PMC passes of tools/profile_gpu.sh):
    # wave64 VALU instructions x 4 issue cycles per SIMD against the kernel's cycles
    vpath = os.path.join(ROOT, "profiles", "valu.json")
    valu = json.load(open(vpath)).get(dom) if os.path.exists(vpath) else None
    out = {
        "metric": "ORB frames/s (752x480, 8 lvl, 1000 kp)",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[1]: synthetic batch of 752x480 uint8 frames, ORB extract+describe, "
                               "nFeatures=1000, 8 levels, scale 1.2, FAST 20/7",
                   "frames_per_step_per_gpu": B, "distinct_frames_per_gpu": unique,
                   "mean_keypoints_per_frame": round(kp_mean, 1), "inputs": "resident in HBM",
                   "contexts_per_gpu": nctx, "blur_taps": "CV_4_0_0", "quadtree_tie_break": "creation order"},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": dom_bytes,
                     "kernel_ms": round(acc[dom], 4),
                     "valu_issue": valu,
                     "stage_ms": {k: round(v, 4) for k, v in acc.items()},
                     "path_bytes_per_frame": BYTES_PER_FRAME,
                     "path_achieved": round(BYTES_PER_FRAME * fps / world / 1e9, 2),
                     "path_frac": round(BYTES_PER_FRAME * fps / world / 1e9 / HBM_PEAK_GBS, 5)},
    }
    if valu:
        # The kernel's real limiter: wave64 VALU instructions (PMC count for this workload, profiles/valu.json)
        # over the live kernel time, against 256 CUs x 4 SIMDs issuing one instruction per 4 cycles at 2.4 GHz.
        issue_peak = 256 * 4 * 2.4e9 / 4
        issue_rate = valu["valu_insts_per_launch"] / (acc[dom] * 1e-3)
        out["roofline_valu"] = {"bound": "valu-issue", "kernel": dom, "achieved": round(issue_rate / 1e9, 2),
                                "peak": round(issue_peak / 1e9, 2), "unit": "G wave64-instructions/s",
                                "frac": round(issue_rate / issue_peak, 4),
                                "frac_of_kernel_cycles": valu["valu_issue_frac"]}
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.cpu_seconds, host[:64])
        out["gpu_over_cpu_all_cores"] = round(fps / out["cpu_baseline"]["value"], 1)
    if world == 1 and not a.no_hamming:
        h = hamming_bench(torch, local)
        out["hamming"] = {k: (round(v, 4) if isinstance(v, float) and v < 1e6 else v) for k, v in h.items()}
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    if use_dist:
        os.dup2(2, 1)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
