#!/bin/bash
# rehearsal of the strong-scaling points on one GPU (virtual ranks): unsplit against adaptive tile splitting
set -e
out=gpurun_out/split
mkdir -p $out
: > $out/summary.txt
run() { echo "== $1 : $2" >> $out/summary.txt; env $1 timeout -k 10 200 python3 profiles/tools/split_calibration.py $2 >> $out/summary.txt 2>> $out/stderr.txt; tail -1 $out/summary.txt; }
run "VPT_SPLIT=1 VPT_SPLIT_VERBOSE=1" "1 256"
for n in 2 4 8; do for r in 0 $((n-1)); do run "VPT_SPLIT=0" "$n 256 $r"; run "VPT_SPLIT_VERBOSE=1" "$n 256 $r"; done; done
run "VPT_SPLIT_VERBOSE=1" "8 256 4"
cat $out/stderr.txt | tail -30
