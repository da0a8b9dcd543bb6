set -e
out=gpurun_out/lean
mkdir -p $out
B="python3 bench.py --steps 3 --warmup 2 --no-cold --cpu-sample 0 --balance"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('balance'))"; }
timeout -k 10 200 $B | show base | tee $out/summary.txt
VPT_HIP_LIB=libvpt_hip_lean.so timeout -k 10 200 $B | show lean | tee -a $out/summary.txt
H="--scene tests/golden/scenes/05_head1ss_sub/head1ss_sub.json --resolution 1280 --spp 64"
timeout -k 10 200 $B $H | show head-base | tee -a $out/summary.txt
VPT_HIP_LIB=libvpt_hip_lean.so timeout -k 10 200 $B $H | show head-lean | tee -a $out/summary.txt
VPT_HIP_LIB=libvpt_hip_lean.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reference_fixtures or head_vol or surf_" 2>&1 | tail -2 | tee -a $out/summary.txt
