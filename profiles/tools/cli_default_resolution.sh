set -e
out=gpurun_out/cli720
mkdir -p $out
S=tests/golden/scenes/03_volume/volume.json
B=volumetric-path-tracer_amd/ypathtrace
VPT_SPLIT=0 $B --scene $S --shader volpathtrace --bounces 64 --output $out/unsplit.png > $out/unsplit.txt 2>&1
VPT_SPLIT_VERBOSE=1 $B --scene $S --shader volpathtrace --bounces 64 --output $out/split.png > $out/split.txt 2>&1
cat $out/unsplit.txt $out/split.txt
cmp $out/unsplit.png $out/split.png && echo "same PNG"
timeout -k 10 300 python3 -m pytest tests/test_tile_splitting.py tests/test_cli.py -x -q -m gpu 2>&1 | tail -3
