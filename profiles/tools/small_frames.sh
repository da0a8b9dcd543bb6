set -e
out=gpurun_out/small
mkdir -p $out
: > $out/summary.txt
for res in 320 640 720 960; do for m in 0 auto; do
  if [ $m = 0 ]; then export VPT_SPLIT=0; else unset VPT_SPLIT; fi
  echo "== resolution $res VPT_SPLIT=$m" >> $out/summary.txt
  timeout -k 10 200 python3 bench.py --resolution $res --steps 3 --warmup 3 --no-cold --cpu-sample 0 --balance 2>>$out/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['config']['workload'], d['value'], d['roofline']['kernel_ms'], d.get('balance'))" >> $out/summary.txt
done; done
cat $out/summary.txt
