#!/bin/bash
# fused Winograd data-gradient + weight-gradient launch: launch time and step time against the number of weight-gradient splits
O=gpurun_out/wino_splits.txt; : > $O
cfgs=("--bs 16" "--blocks 9 --lambda-rs 1 --bs 32" "--padding 10" "--inject --size 512 --padding 10 --bs 8")
for c in "${cfgs[@]}"; do
  for s in auto 3 5 7 9 11 13 15 19 23; do
    if [ $s = auto ]; then unset NIRGAN_WINO_SPLITS; else export NIRGAN_WINO_SPLITS=$s; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 $c 2>/dev/null | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$c | splits $s | step %.2f ms | %s %.1f us  frac %.3f' % (d['ms_per_step'], r['kernel'], r['avg_launch_ms'] * 1e3, r['frac']))" >> $O || exit 1
  done
done
cat $O
