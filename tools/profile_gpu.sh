#!/bin/bash
# Runs on the GPU box: kernel-trace stats pass + PMC passes of a short bench.py run.
#   bash tools/profile_gpu.sh <tag>      -> gpurun_out/prof_<tag>/{stats,pmc_*}/...
set -e -o pipefail
TAG=${1:-x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 200 --warmup 20 --launches 1 --unique 256 --contexts 1 --no-cpu-baseline --min-seconds 0 --legs="
cd /tmp
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 $ARGS > "$OUT/stats_bench.json"
pmc() {
    name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 $ARGS > /dev/null
}
pmc sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
pmc sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS
pmc sq3 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc grbm GRBM_GUI_ACTIVE
cd "$ROOT"
python3 tools/pmc_summary.py $(find "$OUT" -name '*counter_collection.csv') > "$OUT/pmc_summary.txt"
find "$OUT" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
ls "$OUT"
