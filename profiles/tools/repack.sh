# EXPERIMENT: waves of pixels of like cost (VPT_REPACK=1; libraries built with -DVPT_LANE_COST:
#   make variant NAME=costk1 VFLAGS="-DVPT_LANE_COST -DVPT_EXPERIMENT_ONLY_VOLPATH"; make variant NAME=costk2 VFLAGS="-DVPT_LANE_COST -DVPT_EXPERIMENT_ONLY_K2")
# VPT_REPACK_COARSE=c: cost classes of 2^c trips (neighbours stay together within a class); VPT_REPACK_BLOCK=b: sort within blocks of b x b tiles only
set -e
out=gpurun_out/repack
mkdir -p $out
export VPT_SPLIT_VERBOSE=1
B="python3 bench.py --steps 4 --warmup 3 --no-cold --no-others --cpu-sample 0"
K2="--scene tests/golden/scenes/06_gridsdf_full/gridsdf_full.json --shader implicit --bounces 4 --spp 128"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
export VPT_HIP_LIB=variants/libvpt_hip_costk2.so
timeout -k 10 200 $B $K2 2>>$out/err.txt | show k2-tiles | tee -a $out/summary.txt
for b in 2 4; do VPT_REPACK=1 VPT_REPACK_BLOCK=$b timeout -k 10 200 $B $K2 2>>$out/err.txt | show k2-repack-block$b | tee -a $out/summary.txt; done
VPT_REPACK=1 VPT_REPACK_BLOCK=2 VPT_REPACK_COARSE=5 timeout -k 10 200 $B $K2 2>>$out/err.txt | show k2-repack-block2-coarse5 | tee -a $out/summary.txt
export VPT_HIP_LIB=variants/libvpt_hip_costk1.so
timeout -k 10 200 $B 2>>$out/err.txt | show k1-tiles | tee -a $out/summary.txt
for b in 2 4; do VPT_REPACK=1 VPT_SPLIT=1 VPT_REPACK_BLOCK=$b timeout -k 10 200 $B 2>>$out/err.txt | show k1-repack-block$b | tee -a $out/summary.txt; done
VPT_REPACK=1 VPT_SPLIT=1 VPT_REPACK_BLOCK=2 VPT_REPACK_COARSE=4 timeout -k 10 200 $B 2>>$out/err.txt | show k1-repack-block2-coarse4 | tee -a $out/summary.txt
