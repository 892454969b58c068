#!/bin/bash
# Runs on the GPU box: kernel-trace stats + PMC passes of the Hamming top-2 search (65536 x 4M on one GPU).
#   bash tools/profile_knn.sh <tag>      -> gpurun_out/prof_knn_<tag>/...
set -e -o pipefail
TAG=${1:-x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_knn_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="$ROOT/tools/bench_configs.py knn"
cd /tmp
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 $ARGS > "$OUT/bench.json"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -o pmc -- python3 $ARGS > /dev/null
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_lds" -o pmc -- python3 $ARGS > /dev/null
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_grbm" -o pmc -- python3 $ARGS > /dev/null
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(out + "/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if "knn2_kernel" in r["Kernel_Name"]:
            acc["knn2_kernel"][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    lines.append("%s  dispatches=%d" % (k, max(len(v) for v in c.values())))
    for n in sorted(m):
        lines.append("    %-24s %18.1f" % (n, m[n]))
    pairs = 65536.0 * 4194304.0
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    tiles = pairs / 1024
    lines.append("    matrix instructions per 32 x 32 tile of pairs                   %.2f" % (m["SQ_INSTS_MFMA"] / tiles))
    lines.append("    other VALU instructions per tile                                %.2f" % ((m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / tiles))
    lines.append("    matrix pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles) %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc))
    lines.append("    LDS busy (SQ_LDS_IDX_ACTIVE / 256 CUs / cycles)                  %.3f" % (m.get("SQ_LDS_IDX_ACTIVE", 0.0) / 256 / cyc))
open(out + "/knn_pmc_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find "$OUT" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
