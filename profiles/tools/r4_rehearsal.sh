#!/bin/bash
# round 4: virtual ranks of N-GPU jobs on one GPU with the round-4 kernels (as profiles/r02_strong_scaling_rehearsal.txt / r02_config5_3840_rehearsal.txt)
out=gpurun_out/r4_rehearsal
mkdir -p $out
: > $out/summary.txt
run() { echo "== $1" >> $out/summary.txt; timeout -k 10 300 python3 profiles/tools/split_calibration.py $1 >> $out/summary.txt 2>> $out/stderr.txt; tail -1 $out/summary.txt; }
for cfg in "1 256 0" "2 256 0" "2 256 1" "4 256 0" "4 256 3" "8 256 0" "8 256 4" "8 256 7"; do run "$cfg"; done
S="03_volume/volume.json volpathtrace 64 3840"
for cfg in "1 256 0" "2 256 1" "4 256 1" "8 256 0" "8 256 3" "8 256 7"; do run "$cfg $S"; done
