#!/bin/bash
# Development aid: A/B timing of library variants on one GPU box, alternating so that drift hits both alike.
#   tools/ab/liborbx_<name>.so ...   built beforehand;   bash tools/ab_bench.sh A B [rounds]
set -e
ROOT=$(pwd)
R=${3:-3}
for i in $(seq 1 $R); do
  for v in $1 $2; do
    cp "$ROOT/tools/ab/liborbx_$v.so" "$ROOT/my_orb_slam2_amd/liborbx.so"
    timeout -k 10 200 python bench.py --legs "" --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v',round(d['value']),d['roofline']['stage_ms'])"
  done
done
