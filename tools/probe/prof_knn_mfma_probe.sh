#!/bin/bash
# Runs on the GPU box: counters of the matrix-pipe scan at 64 Ki x 4 M (clock, matrix-pipe busy cycles, instruction mix).
set -e -o pipefail
ROOT=$(pwd)
MODE=${1:-2}
OUT=$ROOT/gpurun_out/knn_mfma/prof_$MODE
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="$MODE 65536 4000000 524288 0"
timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- "$ROOT/tools/probe/knn_mfma_probe" $ARGS > "$OUT/run.json"
timeout -s KILL 100 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F6F4 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc1" -o pmc -- "$ROOT/tools/probe/knn_mfma_probe" $ARGS > /dev/null
timeout -s KILL 100 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAVES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA \
    --output-format csv -d "$OUT/pmc2" -o pmc -- "$ROOT/tools/probe/knn_mfma_probe" $ARGS > /dev/null
cd "$ROOT"
python3 tools/pmc_summary.py $(find "$OUT" -name '*counter_collection.csv') > "$OUT/pmc_summary.txt"
find "$OUT" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
cat "$OUT/run.json" "$OUT/pmc_summary.txt"; head -4 "$OUT/kernel_stats.csv"
