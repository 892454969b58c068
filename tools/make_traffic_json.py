#!/usr/bin/env python3
"""Builds profiles/traffic.json from two rocprofv3 PMC passes (one with FETCH_SIZE, one with WRITE_SIZE)
of `bench.py --steps 2 --warmup 1 --unique 256 --no-cpu-baseline --no-hamming`.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950
FETCH_SIZE tallies 128-byte read requests at 64 bytes (TCC_EA0_RDREQ x 64 B), so the read side is
doubled; WRITE_SIZE is exact.  The pyramid entry is the sum over its resize launches of one step.

    python tools/make_traffic_json.py <fetch counter_collection.csv> <write counter_collection.csv>
"""
import csv
import json
import os
import sys
from collections import defaultdict

STAGE = {"pyr_resize": "pyramid", "fast_cells": "fast_cells", "octree": "octree", "orient_desc": "orient_desc"}
LAUNCHES_PER_STEP = {"pyramid": 7, "fast_cells": 1, "octree": 1, "orient_desc": 1}


def mean_per_kernel(path, counter):
    acc = defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        for key, stage in STAGE.items():
            if key in row["Kernel_Name"]:
                acc[stage].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = mean_per_kernel(sys.argv[1], "FETCH_SIZE")
write = mean_per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
raw = {}
for stage in LAUNCHES_PER_STEP:
    n = LAUNCHES_PER_STEP[stage]
    out[stage] = int((2 * fetch[stage] + write[stage]) * 1024 * n)
    raw[stage] = {"FETCH_SIZE_KiB_per_launch": round(fetch[stage], 1), "WRITE_SIZE_KiB_per_launch": round(write[stage], 1),
                  "launches_per_step": n}
out["_raw"] = raw
out["_note"] = "bytes per step of 256 frames (752x480); read side doubled per the gfx950 FETCH_SIZE calibration"
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
