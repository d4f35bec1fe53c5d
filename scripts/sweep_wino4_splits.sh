for s in auto 2 3 4 6 8 12 16; do
  if [ $s = auto ]; then unset NIRGAN_WINO4_SPLITS; else export NIRGAN_WINO4_SPLITS=$s; fi
  timeout -k 10 200 python3 scripts/profile_ops.py 2>/dev/null | grep -E "wino4-dgrad|sum of" | tr '\n' ' '; echo " [splits $s]"
done
for e in 0 1; do NIRGAN_NO_WINOGRAD_DGRAD_ONLY=$e timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probe 2>/dev/null | grep "^{" | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('NO_DGRAD_ONLY=$e', d['value'], d['ms_per_step'])"; done
