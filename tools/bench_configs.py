#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json on one rank's GPU (bench.py stays on configs[1]):

  stereo   configs[2]: pairs of 752x480 frames, nFeatures 1200: left + right extraction and the whole of
           Frame::ComputeStereoMatches on the device -> stereo pairs/s
  hd       configs[3]: 1920x1080 frames, nFeatures 2000 -> frames/s
  knn      configs[4] scaled to one GPU's share: 65536 queries x 4 M database rows per rank, database sharded
           over the ranks, one all-gather + exact merge (RCCL when launched with torchrun) -> pairs/s

    python tools/bench_configs.py [stereo] [hd] [knn]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_configs.py knn

Prints one JSON object per workload (rank 0); every rank works on its own frames / database shard.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(torch, dist, world, fn, steps, warmup):
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def main():
    import torch
    import torch.distributed as dist
    import my_orb_slam2_amd as m
    from my_orb_slam2_amd.sharding import ShardedHammingMatcher

    which = [a for a in sys.argv[1:] if not a.startswith("-")] or ["stereo", "hd", "knn"]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)

    def ensure_dist():
        # The process group comes up after a workload has created its contexts: their streams get hardware queues
        # of their own (four per process on the box), RCCL's idle streams share.
        if world > 1 and not dist.is_initialized():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    if "stereo" in which:
        B, W, H, NF, SETS = 128, 752, 480, 1200, 2          # two sets of (left, right, matcher) in flight
        left = torch.from_numpy(m.synth_batch(rank * B, B, W, H)).cuda()
        right = torch.from_numpy(m.synth_batch(rank * B, B, W, H, right_view=True)).cuda()
        bf, b = 47.90639384423901, 47.90639384423901 / 435.2046959714599     # EuRoC stereo: Camera.bf, bf / fx
        sets = []
        for _ in range(SETS):
            exl = m.ORBextractor(NF, 1.2, 8, 20, 7, device=local, max_width=W, max_height=H, max_batch=B)
            exr = m.ORBextractor(NF, 1.2, 8, 20, 7, device=local, max_width=W, max_height=H, max_batch=B)
            cap = exl.capacity

            def alloc():
                return (torch.empty((B, cap, 7), dtype=torch.float32, device="cuda"),
                        torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda"),
                        torch.zeros(B, dtype=torch.int32, device="cuda"))
            sets.append(dict(exl=exl, exr=exr, L=alloc(), R=alloc(), hm=m.HammingMatcher(local),
                             ur=torch.empty((B, cap), dtype=torch.float32, device="cuda"),
                             z=torch.empty((B, cap), dtype=torch.float32, device="cuda")))

        def step(i):
            s = sets[i % SETS]
            s["exl"].extract_batch_device(left, *s["L"])
            s["exr"].extract_batch_device(right, *s["R"])
            # no host synchronisation: the call orders the three streams on the device (events)
            s["hm"].compute_stereo_matches_device(s["exl"], s["exr"], *s["L"], *s["R"], bf, b, s["ur"], s["z"])
        steps = 48
        ensure_dist()
        dt = timed(torch, dist, world, step, steps, 6)
        for s in sets:
            s["hm"].synchronize()
        if rank == 0:
            matched = float((sets[0]["ur"] >= 0).sum().item()) / B
            print(json.dumps({"workload": "configs[2] stereo pairs, 752x480, nFeatures 1200, extraction x2 + ComputeStereoMatches",
                              "value": round(world * B * steps / dt, 1), "unit": "stereo pairs/s", "n_gpus": world,
                              "pairs_per_step_per_gpu": B, "ms_per_step": round(dt / steps * 1e3, 3),
                              "mean_stereo_matches_per_pair": round(matched, 1), "collective": "none"}))
        del sets

    if "hd" in which:
        B, W, H, NF, NCTX = 64, 1920, 1080, 2000, 3        # three contexts in flight, like bench.py
        ex = [m.ORBextractor(NF, 1.2, 8, 20, 7, device=local, max_width=W, max_height=H, max_batch=B) for _ in range(NCTX)]
        cap = ex[0].capacity
        frames = torch.from_numpy(m.synth_batch(rank * B, B, W, H)).cuda()
        outs = [(torch.empty((B, cap, 7), dtype=torch.float32, device="cuda"),
                 torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda"),
                 torch.zeros(B, dtype=torch.int32, device="cuda")) for _ in range(NCTX)]

        def step(i):
            k, d, c = outs[i % NCTX]
            ex[i % NCTX].extract_batch_device(frames, k, d, c)
        steps = 48
        ensure_dist()
        dt = timed(torch, dist, world, step, steps, 6)
        if rank == 0:
            print(json.dumps({"workload": "configs[3] 1920x1080, nFeatures 2000, 8 levels", "n_gpus": world,
                              "value": round(world * B * steps / dt, 1), "unit": "frames/s",
                              "frames_per_step_per_gpu": B, "ms_per_step": round(dt / steps * 1e3, 3),
                              "mean_keypoints_per_frame": round(float(outs[0][2].float().mean().item()), 1),
                              "collective": "none"}))
        del ex

    if "knn" in which:
        nq, shard = 65536, 4 * 1024 * 1024
        g = torch.Generator(device="cuda").manual_seed(1234 + rank)
        db = torch.randint(0, 256, (shard, 32), dtype=torch.uint8, device="cuda", generator=g)
        gq = torch.Generator(device="cuda").manual_seed(99)      # the same queries on every rank
        q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device="cuda", generator=gq)
        sm = ShardedHammingMatcher(db, rank * shard, device=local)
        ensure_dist()
        res = []

        def step(i):
            res[:] = [sm.knn2(q)]
        steps = 3
        dt = timed(torch, dist, world, step, steps, 1)
        if rank == 0:
            print(json.dumps({"workload": "configs[4] scaled: 65536 queries x %d database rows (%d per GPU), top-2 + merge"
                                          % (world * shard, shard), "n_gpus": world,
                              "value": nq * world * shard * steps / dt, "unit": "Hamming pairs/s",
                              "ms_per_step": round(dt / steps * 1e3, 3),
                              "collective": "one all_gather of nq x 16 B per rank" if world > 1 else "none"}))
    if world > 1 and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
