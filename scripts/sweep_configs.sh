mkdir -p gpurun_out
set -o pipefail
O=gpurun_out/sweep_final.jsonl; : > $O
run() { timeout -k 10 280 python3 bench.py --no-cpu-baseline --sustain 3 "$@" 2> gpurun_out/sweep_err.log | grep '^{' >> $O; echo "done $*"; }
run --blocks 9 --lambda-rs 1 --bs 32 &&
run --inject --size 512 --padding 10 --bs 8 &&
run --padding 10 &&
run --mixed --blocks 9 --lambda-rs 1 --precision bf16 &&
run --mixed --blocks 9 --lambda-rs 1 &&
run --size 128 --bs 64 &&
run --size 512 --bs 4 &&
run --micro 2 &&
run --precision bf16 &&
run --precision bf16x3 &&
timeout -k 10 280 python3 scripts/bench_next_rows.py > gpurun_out/next_rows_final.txt 2>&1; tail -3 gpurun_out/next_rows_final.txt
