set -e
out=gpurun_out/k2feat
mkdir -p $out
S6=tests/golden/scenes/06_gridsdf_synth/gridsdf_synth.json
S7=tests/golden/scenes/07_sdfunction_synth/sdfunction_synth.json
timeout -k 5 90 python3 tests/render_state.py $S6 implicit 64 2 4 $out/sanity.npz > $out/sanity.log 2>&1 || { echo "sanity render failed"; tail -5 $out/sanity.log; exit 1; }
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['kernel_ms'], d.get('balance'))"; }
K2="--scene $S6 --shader implicit --bounces 4 --spp 128 --cpu-sample 0 --steps 3 --warmup 2 --balance --no-cold"
VPT_NO_LEAN=1 timeout -k 10 200 python3 bench.py $K2 2>/dev/null | show 06_general | tee $out/summary.txt
timeout -k 10 200 python3 bench.py $K2 2>/dev/null | show 06_lean | tee -a $out/summary.txt
K7="--scene $S7 --shader implicit --bounces 6 --spp 64 --cpu-sample 0 --steps 3 --warmup 2 --no-cold"
VPT_NO_LEAN=1 timeout -k 10 200 python3 bench.py $K7 2>/dev/null | show 07_general | tee -a $out/summary.txt
timeout -k 10 200 python3 bench.py $K7 2>/dev/null | show 07_lean | tee -a $out/summary.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee -a $out/summary.txt
