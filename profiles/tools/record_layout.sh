# leaf / attribute record layouts A/B: variants built with `make variant NAME=<v> VFLAGS="-DVPT_EXPERIMENT_ONLY_VOLPATH ..."`;
# usage: bash profiles/tools/record_layout.sh base slots     (headline scene and config 3's, two rounds each, then the head tests on the last variant)
set -e
out=gpurun_out/layout
mkdir -p $out
B="python3 bench.py --steps 4 --warmup 2 --no-cold --no-others --cpu-sample 0"
H="--scene tests/golden/scenes/05_head1ss_sub/head1ss_sub.json --resolution 1280 --spp 64"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for round in 1 2; do
  for v in "$@"; do
    VPT_HIP_LIB=variants/libvpt_hip_$v.so timeout -k 10 200 $B | show volume-$v | tee -a $out/summary.txt
    VPT_HIP_LIB=variants/libvpt_hip_$v.so timeout -k 10 200 $B $H | show head-$v | tee -a $out/summary.txt
  done
done
last="${@: -1}"
VPT_HIP_LIB=variants/libvpt_hip_$last.so timeout -k 10 500 python3 -m pytest tests -x -q -m gpu -k "(reference_fixtures and vol) or head_vol_96_4 or surf_ or subdiv_ or intersect_ or surface_ or lights_pdf_ or edge_case_rays or nan_hits or config3_05_head_1280" 2>&1 | tail -3 | tee -a $out/summary.txt
