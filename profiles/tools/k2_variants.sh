# K2 tuning sweep (run inside gpurun): a sanity render under a short timeout first — a kernel that does not come back
# must not take the whole call with it — then the parity tests that touch K2, then the bench per setting.
set -o pipefail
mkdir -p gpurun_out/r2f
S6=tests/golden/scenes/06_gridsdf_synth/gridsdf_synth.json
timeout -k 5 90 python tests/render_state.py $S6 implicit 64 2 4 gpurun_out/r2f/sanity.npz > gpurun_out/r2f/sanity.log 2>&1 || { echo "sanity render failed or timed out"; tail -5 gpurun_out/r2f/sanity.log; exit 1; }
echo "sanity ok"
timeout -k 10 600 python -u -m pytest tests -m gpu -q -x -k "sdf or named or full_size or lights_pdf" > gpurun_out/r2f/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2f/tests.log
K2="--scene $S6 --shader implicit --bounces 4 --spp 128 --cpu-sample 0 --steps 3 --warmup 2"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['kernel_ms'])"; }
timeout -k 10 120 python bench.py $K2 | show default
for g in 0 4 8 12 16 24; do VPT_K2_GRID=$g timeout -k 10 120 python bench.py $K2 | show grid$g; done
timeout -k 10 120 python bench.py $K2 --warmup 0 --steps 1 | show cold_first_launch
K7="--scene tests/golden/scenes/07_sdfunction_synth/sdfunction_synth.json --shader implicit --bounces 6 --spp 64 --cpu-sample 0 --steps 2 --warmup 2"
timeout -k 10 120 python bench.py $K7 | show 07_default
