#!/bin/bash
# Runs on the GPU box: correctness of the matrix-pipe scans against a CPU scan, then their rate.
set -e -o pipefail
P=tools/probe/knn_mfma_probe
OUT=gpurun_out/knn_mfma
mkdir -p $OUT
MODES=${MODES:-"9 10 11"}
: > $OUT/check.jsonl
for mode in $MODES; do
  timeout -k 10 60 $P $mode 1000 5000 4194304 1 >> $OUT/check.jsonl
  timeout -k 10 60 $P $mode 777 3001 1000 1 >> $OUT/check.jsonl
  timeout -k 10 120 $P $mode 2500 40000 16384 1 >> $OUT/check.jsonl
done
cat $OUT/check.jsonl
: > $OUT/rate.jsonl
for mode in $MODES; do
  for chunk in 262144 524288; do
    timeout -k 10 120 $P $mode 65536 4000000 $chunk 0 >> $OUT/rate.jsonl
  done
  timeout -k 10 120 $P $mode 1000 1200 4194304 0 >> $OUT/rate.jsonl
  timeout -k 10 120 $P $mode 2000 100000 8192 0 >> $OUT/rate.jsonl
done
cat $OUT/rate.jsonl
