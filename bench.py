#!/usr/bin/env python3
"""Throughput of the ORB front end on MI355X.

Headline workload (BASELINE.json configs[1]): a synthetic batch of 4096 752x480 uint8 frames, ORB
extract+describe with nFeatures 1000, 8 levels, scale 1.2, FAST 20/7.  One "step" = one pass over that batch
(16 batched launches of 256 frames each, 1024 distinct frames cycled) that already sits in HBM, through the whole
hot path (pyramid, FAST cells, quadtree, orientation + BRIEF); outputs stay on the device.  N > 1: one process per
GPU, every rank extracts its own frames, no data-path collective ("scaling": "weak").

    python bench.py --gpus N --steps K --warmup W       # N > 1: starts N rank processes itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `roofline` is for the kernel that takes most of a step, timed with HIP events on
the library's launch stream; `cpu_baseline` times the CPU oracle (a port of the reference's algorithm; the
reference itself needs OpenCV 4 and cannot be built here) on the host cores.  The other BASELINE configs ride
along as extra keys of the same line, each timed the same way (barrier + device synchronisation either side,
maximum over ranks): `hd1080` (configs[3], frame-parallel), `stereo` (configs[2], pair-parallel, extraction x2 +
the whole of Frame::ComputeStereoMatches), `hamming` (configs[4]: 1 M queries against a 100 M row database
split over the ranks, ONE all-gather of 16 bytes per query and rank, exact fold, ratio test) and `host_fed` (the
headline workload with frames starting and results ending in pinned host memory, three batches in flight: the
PCIe-inclusive figure, never what `value` reports).
The timed region of the headline is at least one second long: the K steps of `--steps` are repeated in whole
rounds until it is (`rounds_timed`), and `ms_per_step` is the mean over all of them.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (spec)
W, H, NFEAT, NLEV, SCALE, INI, MINTH = 752, 480, 1000, 8, 1.2, 20, 7
# Algorithmic bytes per 752x480 frame: every pyramid level read once (1,117,367) + every derived level
# written once (756,407) + outputs 1000 x (28 + 32).
BYTES_PER_FRAME = 1117367 + 756407 + 60000
# The same split by stage: what each kernel has to move at the very least.
LEVEL_PX = 1117367
DERIVED_PX = 756407


def stage_bytes(level_px, derived_px, last_level_px, nkp):
    """What each kernel has to move at the very least, per frame."""
    return {
        "pyramid": (level_px - last_level_px) + derived_px,     # read levels 0..6, write levels 1..7
        "fast_cells": level_px,                                 # read every level once (+ a few KB of candidates)
        "octree": 4 * 10 * nkp + 4 * nkp,                       # ~10 packed candidates per keypoint in, the selected out
        "orient_desc": nkp * (43 * 43 + 60),                    # one 43x43 patch per keypoint + its outputs
    }


STAGE_BYTES = stage_bytes(LEVEL_PX, DERIVED_PX, 210 * 134, 1000)
# configs[3]: 1920x1080 levels sum to 6,419,321 px, derived levels to 4,345,721, last level 536x301, 2000 keypoints
STAGE_BYTES_HD = stage_bytes(6419321, 4345721, 536 * 301, 2000)
BYTES_PER_FRAME_HD = 6419321 + 4345721 + 120000
STAGE_BYTES_STEREO = stage_bytes(LEVEL_PX, DERIVED_PX, 210 * 134, 1200)
PCIE_PEAK_GBS = 63.0           # PCIe Gen5 x16 (spec)
# FP4 matrix peak (dense): 1024 SIMDs x one 32x32x64 instruction (65,536 multiply-adds) per 32 cycles x 2.4 GHz
MFMA_FP4_PEAK_TFLOPS = 1024 * 2 * 65536 / 32 * 2.4e9 / 1e12
LEGS = ("hd1080", "stereo", "hamming", "host_fed")
LEG_SECONDS = 0.5              # shortest timed region of the hd1080 / stereo / host_fed legs (whole rounds of 48 launches)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="frames per batched launch")
    ap.add_argument("--launches", type=int, default=16, help="batched launches per step (16 x 256 = 4096 frames)")
    ap.add_argument("--unique", type=int, default=1024, help="distinct synthetic frames per GPU (cycled)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--legs", default=",".join(LEGS),
                    help="comma-separated extra workloads out of hd1080,stereo,hamming,host_fed ('' = none)")
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="shortest timed region of the headline: the K steps are repeated in whole rounds until it is reached")
    ap.add_argument("--no-hamming", action="store_true", help="same as leaving hamming out of --legs")
    ap.add_argument("--knn-nq", type=int, default=1_000_000)
    ap.add_argument("--knn-ndb", type=int, default=100_000_000)
    ap.add_argument("--contexts", type=int, default=3,
                    help="extractor contexts (HIP streams) a GPU alternates between; the tail of one launch's "
                         "kernels overlaps the head of the next one's")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="process-group backend; gloo is for rehearsals on a box without one GPU per rank")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses device 0 (needs --backend gloo)")
    ap.add_argument("--rehearse", action="store_true",
                    help="launcher / process-group rehearsal without a GPU: no kernels run, value is null")
    return ap.parse_args()


# ---- launcher: `python bench.py --gpus N` without torchrun -----------------------------------------------------

def launch_ranks(a):
    """Starts a.gpus fresh rank processes (this process has not touched torch or HIP) and waits for them. Rank 0
    writes the result line to this process's stdout. Returns the exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    pending = set(range(a.gpus))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print("bench.py: rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr)
                for o in pending:
                    procs[o].terminate()        # the exact processes started above
        time.sleep(0.05)
    return rc


# ---- CPU baselines ----------------------------------------------------------------------------------------------

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


# ---- one rank ---------------------------------------------------------------------------------------------------

class Rank:
    def __init__(self, a):
        self.a = a
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
        # ORBX_BENCH_FORCE_DIST=1 runs the collective code path with a single rank (rehearsal on a one-GPU box)
        self.use_dist = self.world > 1 or os.environ.get("ORBX_BENCH_FORCE_DIST") == "1"
        self.backend_note = None

    def init_group(self):
        import torch
        import torch.distributed as dist
        if not self.use_dist or dist.is_initialized():
            return
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if self.a.backend == "nccl":
            # RCCL over xGMI is the backend of record. If its communicator does not come up on this node (every rank
            # sees that at the same point: the first collective below), the ranks meet again over gloo on the next
            # port and the result line says so; extraction and stereo use no collective in their data path, and the
            # kNN all-gather of 16 bytes per query then travels through host memory.
            try:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", self.local))
                t = torch.ones(1, device=torch.device("cuda", self.local))
                dist.all_reduce(t)
                torch.cuda.synchronize()
                if int(t.item()) != self.world:
                    raise RuntimeError("all_reduce over RCCL returned %r for %d ranks" % (t.item(), self.world))
                return
            except Exception as e:                          # noqa: BLE001 (any failure of the backend takes this path)
                self.backend_note = "nccl failed (%s: %s); gloo used instead" % (type(e).__name__, str(e)[:200])
                print("bench.py: rank %d: %s" % (self.rank, self.backend_note), file=sys.stderr)
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:                           # noqa: BLE001
                    pass
                self.a.backend = "gloo"
                os.environ["MASTER_PORT"] = str(int(os.environ["MASTER_PORT"]) + 1)
        dist.init_process_group(self.a.backend, rank=self.rank, world_size=self.world)

    def fence(self):
        import torch
        import torch.distributed as dist
        torch.cuda.synchronize()
        if self.use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, dt):
        import torch
        import torch.distributed as dist
        if not self.use_dist:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if self.a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, fn, steps, warmup, min_seconds=None):
        """warmup untimed steps, then exactly `steps` steps between two fences; maximum over ranks, in seconds.
        With min_seconds the `steps` steps are repeated in whole rounds (every rank the same number, taken from
        the slowest rank's first round) until the timed region is that long; returns (seconds, rounds)."""
        for i in range(warmup):
            fn(i)
        rounds = 1
        if min_seconds is not None and min_seconds > 0:
            self.fence()
            t0 = time.perf_counter()
            for i in range(steps):
                fn(i)
            self.fence()
            first = self.max_over_ranks(time.perf_counter() - t0)
            rounds = max(1, int(min_seconds / max(first, 1e-6) + 0.999))
        self.fence()
        t0 = time.perf_counter()
        for i in range(steps * rounds):
            fn(i)
        self.fence()
        self.last_own_seconds = time.perf_counter() - t0
        dt = self.max_over_ranks(self.last_own_seconds)
        return (dt, rounds) if min_seconds is not None else dt

    def gather_own_seconds(self):
        """Every rank's own time for the last timed region (rank order), for the per-rank figures."""
        import torch
        import torch.distributed as dist
        if not self.use_dist:
            return [self.last_own_seconds]
        t = torch.tensor([self.last_own_seconds], dtype=torch.float64, device="cuda" if self.a.backend == "nccl" else "cpu")
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return [float(x.item()) for x in out]


class ExtractLeg:
    """Frame-parallel extraction: `nctx` contexts alternate over batched launches of B frames."""

    def __init__(self, R, width, height, nfeat, B, unique, nctx):
        import torch
        import my_orb_slam2_amd as m
        self.R, self.B, self.nctx = R, B, nctx
        self.unique = max(B, (unique + B - 1) // B * B)
        self.host = m.synth_batch(R.rank * self.unique, self.unique, width, height)
        self.frames = torch.from_numpy(self.host).cuda()
        self.exs = [m.ORBextractor(nfeat, SCALE, NLEV, INI, MINTH, device=R.local, max_width=width,
                                   max_height=height, max_batch=B) for _ in range(nctx)]
        cap = self.exs[0].capacity
        self.outs = [(torch.empty((B, cap, 7), dtype=torch.float32, device="cuda"),
                      torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda"),
                      torch.zeros(B, dtype=torch.int32, device="cuda")) for _ in range(nctx)]
        self.nbatches = self.unique // B

        # The contexts' own streams, named explicitly: inputs and outputs are resident and every timed region starts
        # and ends with a device synchronisation, so a launch needs no ordering against torch's current stream (the
        # binding's default, an event record + wait per call, puts a ~16 us bubble in front of every launch chain).
        self.streams = [ex._torch_stream().cuda_stream for ex in self.exs]

    def launch(self, i):
        b = i % self.nbatches
        c = i % self.nctx
        k, d, n = self.outs[c]
        self.exs[c].extract_batch_device(self.frames[b * self.B:(b + 1) * self.B], k, d, n, stream=self.streams[c])

    def mean_keypoints(self):
        return float(self.outs[0][2].float().mean().item())

    def stage_ms(self, reps=9, overlapped=False):
        """Median per-stage kernel time of a launch of context 0 (HIP events on the library's launch stream).
        alone: nothing else runs beside it. overlapped: the launch sits inside the rotation over all contexts, as in
        the timed region (the other contexts' launches are queued before and after it)."""
        ex = self.exs[0]
        ex.set_profiling(True)
        samples = {}
        for i in range(reps):
            if overlapped and self.nctx > 1:
                for c in range(1, self.nctx):
                    self.launch(self.nctx * (2 * i) + c)
                self.launch(self.nctx * (2 * i + 1))                     # context 0, profiled
                for c in range(1, self.nctx):
                    self.launch(self.nctx * (2 * i + 1) + c)
            else:
                # three launches of the same context run straight ahead of the one that is read (the events of a
                # launch replace those of the one before): a launch that follows an idle GPU meets cooler caches and
                # clocks and read 10 % longer than the profiler's average over a run of launches
                for _ in range(4):
                    self.launch(self.nctx * i)
            for k, v in ex.stage_ms().items():
                samples.setdefault(k, []).append(v)
        ex.set_profiling(False)
        # the median of the launches: one launch that meets a clock change or a preemption must not name the dominant kernel
        return {k: sorted(v)[len(v) // 2] for k, v in samples.items()}


def hbm_roofline(stage, per_frame_bytes, frames):
    """roofline object for the stage that takes longest: its algorithmic bytes per launch over its kernel time."""
    dom = max(stage, key=stage.get)
    nbytes = per_frame_bytes[dom] * frames
    achieved = nbytes / (stage[dom] * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes_per_launch": nbytes,
            "kernel_ms": round(stage[dom], 4), "stage_ms": {k: round(v, 4) for k, v in stage.items()}}


class HostFedLeg:
    """The headline workload fed from the host: frames start and results end in pinned host memory, three slots
    (an extractor context and a stream each) queue copy in -> orbx_extract_batch_device -> copy out, so the
    transfers of one batch overlap the kernels of another. PCIe included; never what `value` reports."""

    def __init__(self, main, slots=3):
        import torch
        self.B, self.slots = main.B, min(slots, len(main.exs))
        self.exs = main.exs[:self.slots]
        cap = self.exs[0].capacity
        self.streams = [torch.cuda.Stream() for _ in range(self.slots)]
        self.h_in = [torch.from_numpy(main.host[:self.B]).clone().pin_memory() for _ in range(self.slots)]
        self.d_in = [torch.empty((self.B, H, W), dtype=torch.uint8, device="cuda") for _ in range(self.slots)]
        self.d_out = main.outs[:self.slots]
        self.h_out = [tuple(torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in self.d_out[0]) for _ in range(self.slots)]
        self.bytes_in = int(self.h_in[0].numel())
        self.bytes_out = int(sum(t.numel() * t.element_size() for t in self.h_out[0]))

    def submit(self, i):
        import torch
        k = i % self.slots
        st = self.streams[k]
        st.synchronize()                                 # the slot's previous batch is back in host memory
        with torch.cuda.stream(st):
            self.d_in[k].copy_(self.h_in[k], non_blocking=True)
            self.exs[k].extract_batch_device(self.d_in[k], *self.d_out[k], stream=st.cuda_stream)
            for h, d in zip(self.h_out[k], self.d_out[k]):
                h.copy_(d, non_blocking=True)


class StereoLeg:
    """configs[2]: left + right extraction (nFeatures 1200) and Frame::ComputeStereoMatches, pair p on rank p mod G."""

    # 256 pairs per launch and three sets in flight, like the headline's 256 frames and three contexts: 151.8 k pairs/s
    # where 128 pairs x 2 sets give 140.8 k, 128 x 3 146.8 k, 256 x 2 147.8 k, 192 x 3 150.9 k, 64 x 3 127.1 k
    def __init__(self, R, B=256, sets=3):
        import torch
        import my_orb_slam2_amd as m
        self.B = B
        self.left = torch.from_numpy(m.synth_batch(R.rank * B, B, W, H)).cuda()
        self.right = torch.from_numpy(m.synth_batch(R.rank * B, B, W, H, right_view=True)).cuda()
        self.bf, self.b = 47.90639384423901, 47.90639384423901 / 435.2046959714599   # EuRoC stereo: Camera.bf, bf / fx
        self.sets = []
        for _ in range(sets):
            exl = m.ORBextractor(1200, SCALE, NLEV, INI, MINTH, device=R.local, max_width=W, max_height=H, max_batch=B)
            exr = m.ORBextractor(1200, SCALE, NLEV, INI, MINTH, device=R.local, max_width=W, max_height=H, max_batch=B)
            cap = exl.capacity

            def alloc():
                return (torch.empty((B, cap, 7), dtype=torch.float32, device="cuda"),
                        torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda"),
                        torch.zeros(B, dtype=torch.int32, device="cuda"))
            hm = m.HammingMatcher(R.local)
            # the three streams of a set, named explicitly (see ExtractLeg): inputs and outputs are resident and the
            # timed region starts and ends with a device synchronisation
            self.sets.append(dict(exl=exl, exr=exr, L=alloc(), R=alloc(), hm=hm,
                                  sl=exl._torch_stream().cuda_stream, sr=exr._torch_stream().cuda_stream,
                                  sm=hm._own_stream().cuda_stream,
                                  ur=torch.empty((B, cap), dtype=torch.float32, device="cuda"),
                                  z=torch.empty((B, cap), dtype=torch.float32, device="cuda")))

    def step(self, i):
        s = self.sets[i % len(self.sets)]
        s["exl"].extract_batch_device(self.left, *s["L"], stream=s["sl"])
        s["exr"].extract_batch_device(self.right, *s["R"], stream=s["sr"])
        # no host synchronisation: the call orders the three streams on the device (events)
        s["hm"].compute_stereo_matches_device(s["exl"], s["exr"], *s["L"], *s["R"], self.bf, self.b, s["ur"], s["z"],
                                              stream=s["sm"])

    def finish(self):
        for s in self.sets:
            s["hm"].synchronize()

    def stage_ms(self, reps=5):
        """Median kernel times of one step, each part alone: the left extraction's four stages (the right one is the
        same work) and the kernels of ComputeStereoMatches (band search, SAD refinement, median cut) between two
        events on the matcher's stream."""
        import torch
        s = self.sets[0]
        s["exl"].set_profiling(True)
        samples = {}
        for _ in range(reps):
            s["exl"].extract_batch_device(self.left, *s["L"])
            s["exr"].extract_batch_device(self.right, *s["R"])
            for k, v in s["exl"].stage_ms().items():
                samples.setdefault(k, []).append(v)
            torch.cuda.synchronize()
            st = s["hm"]._own_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            s["hm"].compute_stereo_matches_device(s["exl"], s["exr"], *s["L"], *s["R"], self.bf, self.b, s["ur"], s["z"])
            e1.record(st)
            s["hm"].synchronize()
            samples.setdefault("stereo_matches", []).append(e0.elapsed_time(e1))
        s["exl"].set_profiling(False)
        return {k: sorted(v)[len(v) // 2] for k, v in samples.items()}


def hamming_leg(R, a):
    """configs[4]: nq queries against an ndb-row database split into contiguous shards, one per rank; every rank
    scans its shard for all queries, ONE all-gather of 16 bytes per query and rank, exact fold, ratio test.
    Size-independent checks: every planted query comes back with its row and distance, the ratio test accepts
    exactly the planted queries, and a few queries are recomputed over the whole database with plain torch ops."""
    import torch
    import torch.distributed as dist
    from my_orb_slam2_amd import sharding, synth
    nq, ndb = a.knn_nq, a.knn_ndb
    dev = torch.device("cuda", R.local)
    lo, hi = sharding.shard_range(ndb, R.rank, R.world)
    db = synth.knn_db_rows(lo, hi - lo, dev)
    q, src, planted = synth.knn_queries(nq, ndb, dev)
    sm = sharding.ShardedHammingMatcher(db, lo, device=R.local)
    acc = torch.empty(nq, dtype=torch.uint8, device=dev)
    res = []

    def step(i):
        idx, d1, d2 = sm.knn2(q, slice_rows=131072)
        sm.m.ratio_test_device(d1, d2, acc, th=50, inclusive=True, ratio=0.6)
        sm.m.after()
        res[:] = [idx, d1, d2]
    sm.knn2(q[:4096])                  # scratch buffers and the communicator come up outside the timed region
    dt = R.timed(step, 1, 0)
    idx, d1, d2 = res
    half = nq // 2
    st = {"planted_queries": half,
          "planted_idx_wrong": int((idx[:half] != src).sum().item()),
          "planted_d1_wrong": int((d1[:half] != planted).sum().item()),
          "accepted": int(acc.sum().item()),
          "accepted_not_planted": int(acc[half:].sum().item()),
          "d1_gt_d2": int((d1 > d2).sum().item())}
    ns = 8
    picks = [(i * (nq // ns) + (i % 2) * half) % nq for i in range(ns)]      # planted and uniform ones alternate
    wrong = 0
    for qi in picks:
        kt = torch.tensor(synth.knn_top2_torch(q[qi], db, lo), dtype=torch.int64)
        if R.use_dist:
            kt = kt.cuda() if a.backend == "nccl" else kt
            allk = [torch.empty_like(kt) for _ in range(R.world)]
            dist.all_gather(allk, kt)
            kt = torch.sort(torch.cat(allk).cpu()).values[:2]
        k1, k2 = int(kt[0]), int(kt[1])
        got = (int(idx[qi].item()), int(d1[qi].item()), int(d2[qi].item()))
        wrong += got != (k1 & ((1 << 40) - 1), k1 >> 40, k2 >> 40)
    st["sampled_queries"], st["sampled_wrong"] = ns, wrong
    bad = (st["planted_idx_wrong"] + st["planted_d1_wrong"] + st["accepted_not_planted"] + st["d1_gt_d2"] + wrong +
           (st["accepted"] != half))
    if bad:
        raise SystemExit("bench.py: kNN property checks failed: %s" % json.dumps(st))
    pairs = nq * ndb / dt
    # Matrix-pipe bound: a pair is a 256-term dot product of FP4 operands (v_mfma_f32_32x32x64_f8f6f4, four per
    # 32 x 32 tile of pairs), 512 flop per pair against the dense FP4 peak; the xor + popcount form it replaced was
    # bound by VALU issue at 256 CUs x 64 lanes x 2.4 GHz / 16 = 2.46e12 pairs/s.
    tflops = pairs * 512 / 1e12
    peak = R.world * MFMA_FP4_PEAK_TFLOPS
    out = {"metric": "Hamming pairs/s (top-2 kNN + ratio test, %d x %d, database sharded over %d GPU%s)"
                     % (nq, ndb, R.world, "" if R.world == 1 else "s"),
           "value": pairs, "seconds": round(dt, 3), "n_gpus": R.world, "scaling": "strong",
           "rows_per_gpu": hi - lo, "collective": "one all_gather of nq x 16 B per rank" if R.world > 1 else "none",
           "roofline": {"bound": "mfma", "kernel": "knn2_kernel", "achieved": round(tflops, 1), "peak": round(peak, 1),
                        "unit": "TFLOP/s", "frac": round(tflops / peak, 4), "traffic": None, "dtype": "fp4 (exact: bits as 0 / +-1)",
                        "flop_per_pair": 512, "time": "whole leg (scan + fold + ratio test), wall clock"},
           "valu_form_peak_pairs_per_s": R.world * 256 * 64 * 2.4e9 / 16,
           "checks": st}
    if R.rank == 0 and R.world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = hamming_cpu_baseline(q[:4096].cpu().numpy(), db[: 1 << 20].cpu().numpy())
    return out


def rehearse(R, a, real_stdout):
    """No GPU: brings the ranks up, runs the collectives the timed path uses and prints a line without a value."""
    import torch
    import torch.distributed as dist
    R.use_dist = True
    R.a.backend = "gloo"
    R.init_group()
    dist.barrier()
    t = torch.tensor([float(R.rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ranks = dist.get_world_size()
    if R.rank == 0:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps({"metric": "ORB frames/s (752x480, 8 lvl, 1000 kp)", "value": None, "unit": "frames/s",
                          "n_gpus": R.world, "ranks_in_group": ranks, "backend": "gloo", "rehearsal": True,
                          "max_rank_seen": int(t.item())}), flush=True)
        os.dup2(2, 1)
    dist.destroy_process_group()


def run_rank(a):
    R = Rank(a)
    # Native libraries write to file descriptor 1 on their own (RCCL and gloo print a banner when their first
    # communicator comes up): until the result line, everything that goes to fd 1 is sent to stderr instead.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if a.rehearse:
        return rehearse(R, a, real_stdout)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the measured path)")
    torch.cuda.set_device(R.local)
    legs = [l for l in a.legs.split(",") if l]
    for l in legs:
        if l not in LEGS:
            raise SystemExit("unknown leg %r" % l)
    if a.no_hamming and "hamming" in legs:
        legs.remove("hamming")
    B = a.batch
    nctx = max(1, a.contexts)

    # ---- every workload's contexts come up before the process group: their streams are created first and get
    # hardware queues of their own (the box runs four per process); RCCL's streams, idle while extraction runs,
    # share ----
    main = ExtractLeg(R, W, H, NFEAT, B, a.unique, nctx)
    # 32 frames of 1920x1080 per launch: about as many pixels as 256 frames of 752x480 need to fill the chip; 64 frames
    # per launch measured 6 % lower, 16 frames 14 % lower
    hd = ExtractLeg(R, 1920, 1080, 2000, 32, 64, nctx) if "hd1080" in legs else None
    # (the stereo leg's nine streams come up when its turn comes, after the 1080p leg's contexts are gone: with them
    # alive beside it the 1080p leg ran 6 % slower, 76.4 k instead of 81.3 k frames/s)
    R.init_group()
    ranks_in_group = dist.get_world_size() if R.use_dist else 1

    # ---- headline: a.steps steps of a.launches batched launches each, in whole rounds until >= a.min_seconds ----
    L = a.launches

    def step(i):
        for j in range(L):
            main.launch(i * L + j)
    for i in range(max(8, 2 * nctx)):                   # clocks and caches up before anything is counted
        main.launch(i)
    dt, rounds = R.timed(step, a.steps, a.warmup, a.min_seconds)
    steps_timed = a.steps * rounds
    own_seconds = R.gather_own_seconds()
    kp_mean = main.mean_keypoints()

    # ---- per-stage kernel time, HIP events on the library's launch stream (outside the timed region):
    # context 0 alone, and context 0 inside the rotation over all contexts as in the timed region ----
    acc = main.stage_ms(9, overlapped=False)
    acc_overlapped = main.stage_ms(9, overlapped=True)

    extra = {}
    if "host_fed" in legs:
        hf = HostFedLeg(main)
        for i in range(hf.slots):
            hf.submit(i)
        fsteps = 48
        fdt, frounds = R.timed(hf.submit, fsteps, 3, LEG_SECONDS)
        fsteps *= frounds
        frames_s = R.world * hf.B * fsteps / fdt
        extra["host_fed"] = {"workload": "configs[1] with frames starting and results ending in pinned host memory: %d slots, "
                                         "each copy in -> orbx_extract_batch_device -> copy out on its own stream" % hf.slots,
                             "value": round(frames_s, 1), "unit": "frames/s", "n_gpus": R.world, "frames_per_launch_per_gpu": hf.B,
                             "launches": fsteps, "scaling": "weak", "ms_per_launch": round(fdt / fsteps * 1e3, 3),
                             "bytes_per_launch": {"in": hf.bytes_in, "out": hf.bytes_out},
                             "h2d_gb_s_per_gpu": round(hf.bytes_in * fsteps / fdt / 1e9, 2),
                             "d2h_gb_s_per_gpu": round(hf.bytes_out * fsteps / fdt / 1e9, 2),
                             "pcie_peak_gb_s": PCIE_PEAK_GBS,
                             "h2d_frac_of_pcie": round(hf.bytes_in * fsteps / fdt / 1e9 / PCIE_PEAK_GBS, 4),
                             "bound": "the host link, not the kernels"}
        del hf
    if hd is not None:
        hsteps = 48
        hdt, hrounds = R.timed(hd.launch, hsteps, 6, LEG_SECONDS)
        hsteps *= hrounds
        hd_fps = R.world * hd.B * hsteps / hdt
        hd_stage = hd.stage_ms(5)
        extra["hd1080"] = {"workload": "configs[3]: synthetic 1920x1080 frames, nFeatures 2000, 8 levels, frame-parallel",
                           "value": round(hd_fps, 1), "unit": "frames/s", "n_gpus": R.world,
                           "frames_per_launch_per_gpu": hd.B, "launches": hsteps, "scaling": "weak",
                           "ms_per_launch": round(hdt / hsteps * 1e3, 3), "collective": "none",
                           "mean_keypoints_per_frame": round(hd.mean_keypoints(), 1),
                           "roofline": dict(hbm_roofline(hd_stage, STAGE_BYTES_HD, hd.B),
                                            path_bytes_per_frame=BYTES_PER_FRAME_HD,
                                            path_frac=round(BYTES_PER_FRAME_HD * hd_fps / R.world / 1e9 / HBM_PEAK_GBS, 5))}
    del hd
    torch.cuda.empty_cache()
    stereo = StereoLeg(R) if "stereo" in legs else None
    if stereo is not None:
        ssteps = 48
        sdt, srounds = R.timed(stereo.step, ssteps, 6, LEG_SECONDS)
        ssteps *= srounds
        stereo.finish()
        matched = float((stereo.sets[0]["ur"] >= 0).sum().item()) / stereo.B
        st_stage = stereo.stage_ms()
        st_bytes = dict(STAGE_BYTES_STEREO, stereo_matches=2 * 1200 * 60 + 1200 * 2 * 11 * 21)   # both sides' keypoints + descriptors, the SAD windows
        extra["stereo"] = {"workload": "configs[2]: stereo pairs 752x480, nFeatures 1200, left + right extraction + "
                                       "Frame::ComputeStereoMatches, pair p on rank p mod G",
                           "value": round(R.world * stereo.B * ssteps / sdt, 1), "unit": "stereo pairs/s",
                           "n_gpus": R.world, "pairs_per_launch_per_gpu": stereo.B, "launches": ssteps,
                           "scaling": "weak", "ms_per_launch": round(sdt / ssteps * 1e3, 3), "collective": "none",
                           "mean_stereo_matches_per_pair": round(matched, 1),
                           "roofline": dict(hbm_roofline(st_stage, st_bytes, stereo.B),
                                            note="stage times of ONE side's extraction (the other side repeats them) and of "
                                                 "the ComputeStereoMatches kernels, each alone")}
    host64 = main.host[:64]
    del main, stereo
    torch.cuda.empty_cache()
    if "hamming" in legs:
        extra["hamming"] = hamming_leg(R, a)

    if R.rank != 0:
        if R.use_dist:
            dist.destroy_process_group()
        return
    world = R.world
    frames_per_step = B * L
    fps = world * frames_per_step * steps_timed / dt
    dom = max(acc, key=acc.get)
    dom_bytes = STAGE_BYTES[dom] * B
    achieved = dom_bytes / (acc[dom] * 1e-3) / 1e9
    # HBM traffic and VALU instruction counts of the same kernels come from rocprofv3 PMC passes of this program run
    # by the builder (tools/profile_gpu.sh); they are static files, not measured by this run, and tagged as such.
    traffic = valu = None
    tpath, vpath = os.path.join(ROOT, "profiles", "traffic.json"), os.path.join(ROOT, "profiles", "valu.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(dom)
    if os.path.exists(vpath):
        valu = json.load(open(vpath)).get(dom)
    pmc_source = "profiles/traffic.json + profiles/valu.json: builder-run rocprofv3 --pmc passes, not measured by this run"
    out = {
        "metric": "ORB frames/s (752x480, 8 lvl, 1000 kp)",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / steps_timed * 1e3, 4), "steps_timed": steps_timed, "rounds_timed": rounds,
        "timed_seconds": round(dt, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[1]: synthetic batch of 4096 752x480 uint8 frames, ORB extract+describe, "
                               "nFeatures=1000, 8 levels, scale 1.2, FAST 20/7",
                   "frames_per_step_per_gpu": frames_per_step, "launches_per_step": L, "frames_per_launch": B,
                   "distinct_frames_per_gpu": a.unique,
                   "mean_keypoints_per_frame": round(kp_mean, 1), "inputs": "resident in HBM",
                   "contexts_per_gpu": nctx, "blur_taps": "CV_4_0_0", "quadtree_tie_break": "creation order",
                   "ranks_in_group": ranks_in_group, "backend": a.backend if R.use_dist else "none",
                   **({"backend_note": R.backend_note} if R.backend_note else {})},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "traffic_source": pmc_source if traffic is not None else None,
                     "algorithmic_bytes_per_launch": dom_bytes,
                     "kernel_ms": round(acc[dom], 4),
                     "valu_issue": valu,
                     "stage_ms": {k: round(v, 4) for k, v in acc.items()},
                     "stage_ms_overlapped": {k: round(v, 4) for k, v in acc_overlapped.items()},
                     "stage_ms_note": "stage_ms: a launch of one context with nothing beside it (kernel_ms and achieved "
                                      "use it); stage_ms_overlapped: the same launch inside the rotation over all "
                                      "contexts, as in the timed region",
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
                                "frac_of_kernel_cycles": valu["valu_issue_frac"],
                                "valu_cycles_per_inst": 4,
                                "valu_cycles_per_inst_source": "probe: tools/probe/valu_issue_probe.hip, "
                                                               "profiles/r01_valu_issue_probe.txt (4.0-4.4 cycles per wave64 "
                                                               "instruction per SIMD at 8 waves per SIMD, every class used here); "
                                                               "SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.008 quad-cycles agrees",
                                "instruction_count_source": pmc_source}
    if world > 1:
        out["per_rank_value"] = [round(frames_per_step * steps_timed / t, 1) for t in own_seconds]
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.cpu_seconds, host64)
        out["gpu_over_cpu_all_cores"] = round(fps / out["cpu_baseline"]["value"], 1)
    for k, v in extra.items():
        out[k] = {kk: (round(vv, 4) if isinstance(vv, float) and vv < 1e6 else vv) for kk, vv in v.items()}
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    if R.use_dist:
        os.dup2(2, 1)
        dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Nothing in this process has touched torch or HIP yet: start the ranks as children and report their fate.
        sys.exit(launch_ranks(a))
    run_rank(a)


if __name__ == "__main__":
    main()
