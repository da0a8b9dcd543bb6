# K1 experiments of round 3 (run inside gpurun): bash profiles/tools/k1_r3.sh <outdir> <variant> [<variant> ...]
# For every variants/libvpt_hip_<variant>.so ("default" = the shipped library): a sanity render under a short timeout, optionally
# (TESTS_FOR="a b") the parity tests that touch K1's volpathtrace instances, then the benches: 03_volume (config 2), the same
# with the LDS part of the stacks capped (VPT_STACK_LDS: the HBM-overflow instance), 05_head1ss_sub (config 3).
set -o pipefail
out=$1; shift
mkdir -p $out
S3=tests/golden/scenes/03_volume/volume.json
S5=tests/golden/scenes/05_head1ss_sub/head1ss_sub.json
B="--cpu-sample 0 --steps 4 --warmup 2 --no-cold --no-others"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for v in "$@"; do
  if [ "$v" = default ]; then unset VPT_HIP_LIB; else export VPT_HIP_LIB=variants/libvpt_hip_$v.so; fi
  timeout -k 5 90 python tests/render_state.py $S3 volpathtrace 64 2 64 $out/sanity_$v.npz > $out/sanity_$v.log 2>&1 || { echo "$v: sanity render failed or timed out"; tail -5 $out/sanity_$v.log; exit 1; }
  if [[ " $TESTS_FOR " == *" $v "* ]]; then
    timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "(fixtures and vol_) or head_vol or surf_subdiv or config2 or config3 or config5 or batching or determinism or instructor_image_lowres" > $out/tests_$v.log 2>&1 || { echo "$v: TESTS FAILED"; tail -15 $out/tests_$v.log; exit 1; }
    echo "$v: $(tail -1 $out/tests_$v.log)"
  fi
  timeout -k 10 200 python bench.py $B 2>$out/err_$v.txt | show "$v 03_volume" || exit 1
  for cap in $STACK_CAPS; do VPT_STACK_LDS=$cap timeout -k 10 200 python bench.py $B 2>>$out/err_$v.txt | show "$v 03_volume LDS$cap" || exit 1; done
  timeout -k 10 200 python bench.py $B --scene $S5 --spp 64 2>>$out/err_$v.txt | show "$v 05_head" || exit 1
  for cap in $HEAD_CAPS; do VPT_STACK_LDS=$cap timeout -k 10 200 python bench.py $B --scene $S5 --spp 64 2>>$out/err_$v.txt | show "$v 05_head LDS$cap" || exit 1; done
done
