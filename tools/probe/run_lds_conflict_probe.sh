#!/bin/bash
# Runs on the GPU box: the LDS bank-conflict probe under rocprofv3 counter passes, one configuration per process.
#   bash tools/probe/run_lds_conflict_probe.sh  -> gpurun_out/ldsprobe/{summary.txt, pmc_*}
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/ldsprobe
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$OUT/counters.txt" 2>&1 || true
: > "$OUT/runs.jsonl"
for cfg in "1 0" "2 0" "4 0" "8 0" "32 0" "1 1"; do
    set -- $cfg
    tag=s$1m$2
    timeout -s KILL 60 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS \
        --output-format csv -d "$OUT/pmc_$tag" -o pmc -- "$ROOT/tools/probe/lds_conflict_probe" $1 $2 >> "$OUT/runs.jsonl"
    if grep -q "SQ_LDS_IDX_ACTIVE" "$OUT/counters.txt"; then
        timeout -s KILL 60 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE \
            --output-format csv -d "$OUT/pmc2_$tag" -o pmc -- "$ROOT/tools/probe/lds_conflict_probe" $1 $2 > /dev/null
    fi
done
cd "$ROOT"
for d in "$OUT"/pmc_* "$OUT"/pmc2_*; do
    [ -d "$d" ] || continue
    echo "== $(basename $d)"
    python3 tools/pmc_summary.py $(find "$d" -name '*counter_collection.csv')
done > "$OUT/summary.txt"
cat "$OUT/runs.jsonl"
