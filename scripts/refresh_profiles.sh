#!/bin/bash
# ONE command that regenerates the judged measurement set of the default bench (run on the GPU box through gpurun):
#     gpurun --timeout 1200 -- 'bash scripts/refresh_profiles.sh r03'
# plain run, rocprofv3 kernel trace + stats of the same command, PMC passes (one counter group per pass: FETCH_SIZE and WRITE_SIZE
# do not fit one pass; never combined with a trace domain), per-op times.  Results land in gpurun_out/refresh/ with the round tag;
# copy them to profiles/ (cp gpurun_out/refresh/<tag>_* profiles/).  The PMC summary is stamped with the hash of the kernel
# sources (scripts/pmc_summary.py::kernel_source_sha16 = bench.py::kernel_source_sha16): bench.py replays its traffic / MFMA-busy
# figures only while that hash matches the sources it is running.
set -o pipefail
TAG=${1:-r04}
export TMPDIR=/tmp
R=$PWD/gpurun_out/refresh; rm -rf $R; mkdir -p $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -- python3 bench.py --no-cpu-baseline --sustain 0 > $R/${TAG}_bench_under_rocprof.json.log 2>&1 || exit 2
echo kernel trace done
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/pmc1 -- python3 bench.py --no-cpu-baseline --no-probe --sustain 0 --steps 3 --warmup 1 > $R/pmc1.log 2>&1 || exit 3
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc2 -- python3 bench.py --no-cpu-baseline --no-probe --sustain 0 --steps 3 --warmup 1 > $R/pmc2.log 2>&1 || exit 4
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pmc3 -- python3 bench.py --no-cpu-baseline --no-probe --sustain 0 --steps 3 --warmup 1 > $R/pmc3.log 2>&1 || exit 5
echo pmc done
python3 scripts/pmc_summary.py $R/pmc1 $R/pmc2 $R/pmc3 > $R/${TAG}_pmc_bench_summary.json || exit 6
# the plain run LAST of the fp32 set, with the summary just recorded in place: its line replays traffic / MFMA busy of THIS build
cp $R/${TAG}_pmc_bench_summary.json profiles/
timeout -k 10 400 python3 bench.py > $R/${TAG}_bench_final.json.log 2>&1 || exit 1
echo bench done
cp $(find $R/kt -name "*kernel_stats.csv" | head -1) $R/${TAG}_bench_kernel_stats.csv || exit 7
timeout -k 10 300 python3 scripts/profile_ops.py > $R/${TAG}_per_op_times.txt 2>&1
# the drop-in API as Lightning drives it, and one rank over RCCL with the data-parallel check (cited by DESIGN.md sections 5 / 6)
timeout -k 10 300 python3 bench.py --no-cpu-baseline --api lightning > $R/${TAG}_bench_api_lightning.json.log 2>&1
NIRGAN_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --verify-dp --sustain 0 > $R/${TAG}_bench_rccl_one_rank.json.log 2>&1
rm -rf $R/kt $R/pmc1 $R/pmc2 $R/pmc3 $R/pmc*.log
# bf16 operand mode (BASELINE.json configs[4]'s arithmetic): the 6-block step under the kernel trace (the dominant bf16 kernel's average
# duration on random operands), the mixed-resolution configs[4] line, per-op times, MFMA-busy counters of the 9-block step
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/ktb -- python3 bench.py --no-cpu-baseline --sustain 0 --precision bf16 > $R/${TAG}_bench_bf16_under_rocprof.json.log 2>&1
cp $(find $R/ktb -name "*kernel_stats.csv" | head -1) $R/${TAG}_bench_bf16_kernel_stats.csv
rm -rf $R/ktb
timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustain 0 --precision bf16 > $R/${TAG}_bench_bf16.json.log 2>&1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustain 0 --mixed --precision bf16 --blocks 9 --lambda-rs 1 > $R/${TAG}_bench_mixed_bf16.json.log 2>&1
timeout -k 10 300 python3 scripts/profile_ops.py 16 6 0 bf16 > $R/${TAG}_per_op_times_bf16.txt 2>&1
timeout -k 10 300 python3 scripts/profile_ops.py 16 9 0 bf16 > $R/${TAG}_per_op_times_bf16_9block.txt 2>&1
A="--no-cpu-baseline --no-probe --sustain 0 --steps 3 --warmup 1 --precision bf16"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/pmcb -- python3 bench.py $A > $R/pmcb.log 2>&1 && python3 scripts/pmc_summary.py $R/pmcb > $R/${TAG}_pmc_bf16_mfma_busy.json
rm -rf $R/pmcb $R/pmcb.log
ls -la $R
