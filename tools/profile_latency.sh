#!/bin/bash
# Runs on the GPU box: average memory latencies per kernel of the extraction launch chain (one context alone), from
# rocprofv3's derived metrics (in-flight instruction levels accumulated / instructions), one pass each.
#   bash tools/profile_latency.sh <tag>      -> gpurun_out/lat_<tag>/summary.txt
set -e -o pipefail
TAG=${1:-x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/lat_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 60 --warmup 10 --launches 1 --unique 256 --contexts 1 --no-cpu-baseline --min-seconds 0 --legs="
cd /tmp
for m in VmemLatency LdsLatency SmemLatency; do
    timeout -s KILL 120 rocprofv3 --pmc $m --output-format csv -d "$OUT/pmc_$m" -o pmc -- python3 $ARGS > /dev/null
done
cd "$ROOT"
python3 tools/pmc_summary.py $(find "$OUT" -name '*counter_collection.csv') > "$OUT/summary.txt"
cat "$OUT/summary.txt"
