# K2 A/B of experiment builds (make variant NAME=<v> VFLAGS="-DVPT_EXPERIMENT_ONLY_K2 ..."): bash profiles/tools/k2_ab.sh k2base k2m
set -e
out=gpurun_out/k2ab
mkdir -p $out
B="python3 bench.py --steps 4 --warmup 2 --no-cold --no-others --cpu-sample 0"
K2="--scene tests/golden/scenes/06_gridsdf_full/gridsdf_full.json --shader implicit --bounces 4 --spp ${K2SPP:-128}"
S7="--scene tests/golden/scenes/07_sdfunction_synth/sdfunction_synth.json --shader implicit --bounces 6 --spp 64"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['kernel_ms'])"; }
for round in ${ROUNDS:-1 2}; do
  for v in "$@"; do
    VPT_HIP_LIB=variants/libvpt_hip_$v.so timeout -k 10 200 $B $K2 | show gridsdf-$v | tee -a $out/summary.txt
    VPT_HIP_LIB=variants/libvpt_hip_$v.so timeout -k 10 200 $B $S7 | show sdfunction-$v | tee -a $out/summary.txt
  done
done
last="${@: -1}"; [ -n "$NOTESTS" ] && exit 0
VPT_HIP_LIB=variants/libvpt_hip_$last.so timeout -k 10 500 python3 -m pytest tests -x -q -m gpu -k "implicit or sdf or spheretrace or volume_sdf or gridsdf" 2>&1 | tail -3 | tee -a $out/summary.txt
