#!/bin/bash
# Development aid: A/B timing of the matcher leg of bench.py for two library variants on one GPU box, alternating.
#   tools/ab/liborbx_<name>.so ...   built beforehand;   bash tools/ab_knn.sh A B [rounds] [nq] [ndb]
set -e
ROOT=$(pwd)
R=${3:-2}
NQ=${4:-1000000}
NDB=${5:-100000000}
for i in $(seq 1 $R); do
  for v in $1 $2; do
    cp "$ROOT/tools/ab/liborbx_$v.so" "$ROOT/my_orb_slam2_amd/liborbx.so"
    timeout -k 10 300 python bench.py --legs hamming --no-cpu-baseline --steps 2 --warmup 1 --min-seconds 0 --knn-nq $NQ --knn-ndb $NDB 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());h=d['hamming'];print('$v','%.4g'%h['value'],h['seconds'],h['checks'])"
  done
done
