#!/bin/bash
# BASELINE config 5 (03_volume 3840x1600) on one GPU: the whole frame, then ranks of an 8-GPU job as virtual ranks
set -e
out=gpurun_out/config5
mkdir -p $out
: > $out/summary.txt
S="03_volume/volume.json volpathtrace 64 3840"
for cfg in "1 256 0" "8 256 0" "8 256 3" "8 256 7" "4 256 1" "2 256 1"; do
  timeout -k 10 400 python3 profiles/tools/split_calibration.py $cfg $S >> $out/summary.txt 2>> $out/stderr.txt; tail -1 $out/summary.txt
done
