#!/usr/bin/env python3
"""Builds profiles/valu.json from the SQ and GRBM PMC passes of tools/profile_gpu.sh.

Per kernel and launch: wave64 VALU instructions, the cycles one SIMD needs to issue its share of them
(4 cycles per instruction, 1024 SIMDs; tools/probe/valu_issue_probe.hip measures the 4), the kernel's
duration in shader cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs), and the ratio of the two; the same
for the LDS: SQ_LDS_IDX_ACTIVE counts the cycles the LDS array of a CU works on indexed operations and
SQ_LDS_BANK_CONFLICT the part of them that bank conflicts add, both in plain cycles summed over the CUs
(tools/probe/lds_conflict_probe.hip: 2.0 / 4.0 / 8.0 active cycles and 0 / 2.0 / 6.0 conflict cycles per
ds_read_b32 at 1 / 2 / 4 lanes per bank; SQ_ACTIVE_INST_LDS is 1.0 per instruction whatever the conflicts,
i.e. an issue count, not a measure of how busy the LDS is).

    python tools/make_valu_json.py gpurun_out/prof_<tag>/pmc_summary.txt
"""
import json
import os
import re
import sys

STAGE = {"pyr_resize": "pyramid", "fast_cells": "fast_cells", "octree": "octree", "orient_desc": "orient_desc"}
LAUNCHES = {"pyramid": 7, "fast_cells": 1, "octree": 1, "orient_desc": 1}
acc, cur = {}, None
for line in open(sys.argv[1]):
    m = re.match(r"(\S.*?)\s+dispatches=", line)
    if m:
        cur = next((s for k, s in STAGE.items() if k in m.group(1)), None)
        continue
    if cur and line.strip():
        k, v = line.split()
        acc.setdefault(cur, {})[k] = float(v)
out = {}
for stage, c in acc.items():
    n = LAUNCHES[stage]
    cycles = c["GRBM_GUI_ACTIVE"] / 8
    valu_cycles = c["SQ_INSTS_VALU"] * 4 / 1024
    lds_cycles = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / 256
    conflict_cycles = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / 256
    out[stage] = {"launches_per_step": n,
                  "valu_insts_per_launch": int(c["SQ_INSTS_VALU"]),
                  "waves_per_launch": int(c["SQ_WAVES"]),
                  "valu_insts_per_wave": round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1),
                  "simd_issue_cycles_per_launch": int(valu_cycles),
                  "kernel_cycles_per_launch": int(cycles),
                  "valu_issue_frac": round(valu_cycles / cycles, 3),
                  "salu_per_valu": round(c.get("SQ_INSTS_SALU", 0.0) / c["SQ_INSTS_VALU"], 3),
                  "lds_busy_frac": round(lds_cycles / cycles, 3),
                  "lds_bank_conflict_frac": round(conflict_cycles / cycles, 3)}
out["_note"] = ("per launch of a 256-frame step (752x480); valu_issue_frac = VALU instructions x 4 cycles / 1024 SIMDs "
                "/ kernel cycles; lds_busy_frac = SQ_LDS_IDX_ACTIVE / 256 CUs / kernel cycles, lds_bank_conflict_frac = "
                "SQ_LDS_BANK_CONFLICT / 256 CUs / kernel cycles (both counters in plain cycles: tools/probe/lds_conflict_probe.hip)")
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "valu.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
