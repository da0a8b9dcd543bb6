# does the rate depend on the samples per launch?  (configs 3 and 4 at several batch sizes)
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
B="python3 bench.py --steps 3 --warmup 1 --no-cold --no-others --cpu-sample 0"
for spp in 64 256 1024; do timeout -k 10 200 $B --scene tests/golden/scenes/05_head1ss_sub/head1ss_sub.json --resolution 1280 --spp $spp | show head-$spp; done
for spp in 128 512; do timeout -k 10 200 $B --scene tests/golden/scenes/06_gridsdf_full/gridsdf_full.json --shader implicit --bounces 4 --spp $spp | show gridsdf-$spp; done
for spp in 64 256 1024; do timeout -k 10 200 $B --spp $spp | show volume-$spp; done
