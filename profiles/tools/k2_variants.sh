# K2 tuning sweep (run inside gpurun): a sanity render under a short timeout first — a kernel that does not come back
# must not take the whole call with it — then the parity tests that touch K2, then the bench per setting.
set -o pipefail
out=gpurun_out/r2n
mkdir -p $out
S6=tests/golden/scenes/06_gridsdf_synth/gridsdf_synth.json
timeout -k 5 90 python tests/render_state.py $S6 implicit 64 2 4 $out/sanity.npz > $out/sanity.log 2>&1 || { echo "sanity render failed or timed out"; tail -5 $out/sanity.log; exit 1; }
echo "sanity ok"
timeout -k 10 600 python -u -m pytest tests -m gpu -q -x -k "sdf or named or lights_pdf or spheretrace or config4" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
K2="--scene $S6 --shader implicit --bounces 4 --spp 128 --cpu-sample 0 --steps 3 --warmup 2 --balance"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['kernel_ms'], d.get('balance'))"; }
timeout -k 10 120 python bench.py $K2 | show default
for v in s24 s24st8 s32st8 w3s24st8; do VPT_HIP_LIB=variants/libvpt_hip_$v.so timeout -k 10 120 python bench.py $K2 | show $v; done
K7="--scene tests/golden/scenes/07_sdfunction_synth/sdfunction_synth.json --shader implicit --bounces 6 --spp 64 --cpu-sample 0 --steps 2 --warmup 2"
timeout -k 10 120 python bench.py $K7 | show 07_default
timeout -k 10 120 python bench.py --steps 3 --warmup 2 --cpu-sample 0 --balance | show 03_volume
