#!/bin/bash
# the other shaders / scenes with the shipped kernels: Msamples/s, kernel ms, fraction of the roofline on each one's own algorithmic bytes
out=gpurun_out/shaders
mkdir -p $out
: > $out/summary.txt
B="python3 bench.py --steps 3 --warmup 2 --no-cold --no-others"
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['algorithmic_bytes_per_sample'])"; }
for sh in pathtrace naive eyelight color; do timeout -k 10 300 $B --shader $sh --cpu-sample 320x16 2>/dev/null | show 03_volume_$sh | tee -a $out/summary.txt; done
timeout -k 10 300 $B --scene tests/golden/scenes/01_surface_min/surface_min.json --shader pathtrace --bounces 4 --cpu-sample 320x16 2>/dev/null | show 01_surface_min_pathtrace_b4 | tee -a $out/summary.txt
timeout -k 10 300 $B --scene tests/golden/scenes/03_volume_lobes/volume_lobes.json --cpu-sample 320x16 2>/dev/null | show 03_volume_lobes_volpathtrace | tee -a $out/summary.txt
timeout -k 10 300 $B --scene tests/golden/scenes/07_sdfunction_synth/sdfunction_synth.json --shader implicit --bounces 6 --spp 64 --cpu-sample 320x16 2>/dev/null | show 07_sdfunction_synth_implicit_b6 | tee -a $out/summary.txt
timeout -k 10 300 $B --scene tests/golden/scenes/08_subdiv_synth/subdiv_synth.json --shader pathtrace --bounces 4 --cpu-sample 320x16 2>/dev/null | show 08_subdiv_synth_pathtrace_b4 | tee -a $out/summary.txt
timeout -k 10 300 $B --scene tests/golden/scenes/06_gridsdf_full/gridsdf_full.json --shader implicit_normal --spp 128 --cpu-sample 320x16 2>/dev/null | show 06_gridsdf_full_implicit_normal | tee -a $out/summary.txt
