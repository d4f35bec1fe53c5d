#!/bin/bash
# A/B of engine options on the GPU box: bash scripts/ab_options.sh "<bench args>" "<NIRGAN_OPTIONS value>" ["<another>" ...]
# prints value / ms_per_step per option set (first line: defaults).
mkdir -p gpurun_out/ab
ARGS=$1; shift
run() { NIRGAN_OPTIONS="$1" timeout -k 10 280 python3 bench.py --no-cpu-baseline --sustain 0 --no-probe $ARGS 2> gpurun_out/ab/err.log | grep '^{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('%-50s %9.2f tiles/s %8.3f ms' % (sys.argv[1] or '(defaults)', j['value'], j['ms_per_step']))" "$1"; }
run "" || exit 1
for o in "$@"; do run "$o" || exit 1; done
