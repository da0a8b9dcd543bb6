# K2 experiments of round 3 (run inside gpurun): bash profiles/tools/k2_r3.sh <outdir> <variant> [<variant> ...]
# For every variants/libvpt_hip_<variant>.so: a sanity render under a short timeout (a kernel that does not come back must
# not take the call with it), the parity tests that touch K2, then the bench of config 4's workload (06_gridsdf_full) and
# of 07_sdfunction_synth.  "default" stands for the shipped libvpt_hip.so.  TESTS_FOR="a b": run the tests for these variants only.
set -o pipefail
out=$1; shift
mkdir -p $out
S6=tests/golden/scenes/06_gridsdf_full/gridsdf_full.json
S7=tests/golden/scenes/07_sdfunction_synth/sdfunction_synth.json
K2="--scene $S6 --shader implicit --bounces 4 --spp 128 --cpu-sample 0 --steps 3 --warmup 2 --balance --no-others"
K7="--scene $S7 --shader implicit --bounces 6 --spp 64 --cpu-sample 0 --steps 2 --warmup 2 --no-cold --no-others"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['kernel_ms'], d.get('balance'), 'cold', (d.get('cold') or {}).get('value'))"; }
for v in "$@"; do
  if [ "$v" = default ]; then unset VPT_HIP_LIB; else export VPT_HIP_LIB=variants/libvpt_hip_$v.so; fi
  timeout -k 5 90 python tests/render_state.py $S6 implicit 64 2 4 $out/sanity_$v.npz > $out/sanity_$v.log 2>&1 || { echo "$v: sanity render failed or timed out"; tail -5 $out/sanity_$v.log; exit 1; }
  if [ -z "$TESTS_FOR" ] || [[ " $TESTS_FOR " == *" $v "* ]]; then
  timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_kat.py tests/test_feature_instances.py -m gpu -q -x -k "sdf or implicit or lights_pdf or spheretrace or config4 or k2" > $out/tests_$v.log 2>&1 || { echo "$v: TESTS FAILED"; tail -15 $out/tests_$v.log; exit 1; }
  echo "$v: $(tail -1 $out/tests_$v.log)"
  fi
  timeout -k 10 200 python bench.py $K2 2>$out/err_$v.txt | show "$v 06_full" || exit 1
  timeout -k 10 200 python bench.py $K7 2>>$out/err_$v.txt | show "$v 07_sdfn" || exit 1
done
