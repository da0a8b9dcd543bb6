#!/bin/bash
# calibration of the tile-splitting model: per-rank time of virtual ranks under different gain tables
set -e
out=gpurun_out/tune
mkdir -p $out
: > $out/summary.txt
run() { echo "== $1 : $2" >> $out/summary.txt; env "$1" timeout -k 10 200 python3 profiles/tools/split_calibration.py $2 >> $out/summary.txt 2>> $out/stderr.txt; tail -1 $out/summary.txt; }
for cfg in "8 256 0" "8 256 7" "2 256 1" "4 256 3"; do
run "VPT_X=0" "$cfg"
run "VPT_SPLIT_TUNE=0.81,0.60,0.41,0.33,0.25,0.17,0.46,0.98" "$cfg"
run "VPT_SPLIT_TUNE=0.81,0.60,0.41,0.33,0.25,0.17,0.30,0.98" "$cfg"
run "VPT_SPLIT_TUNE=0.78,0.56,0.38,0.30,0.22,0.15,0.46,0.98" "$cfg"
run "VPT_SPLIT_TUNE=0.81,0.62,0.45,0.35,0.27,0.20,0.65,0.98" "$cfg"
done
