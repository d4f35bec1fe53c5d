set -o pipefail
export TMPDIR=/tmp
R=$PWD/gpurun_out/pmc_bf16; rm -rf $R; mkdir -p $R
A="--no-cpu-baseline --no-probe --sustain 0 --steps 3 --warmup 1 --precision bf16 --blocks 9"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/pmc1 -- python3 bench.py $A > $R/pmc1.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc2 -- python3 bench.py $A > $R/pmc2.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pmc3 -- python3 bench.py $A > $R/pmc3.log 2>&1 || exit 5
python3 scripts/pmc_summary.py $R/pmc1 $R/pmc2 $R/pmc3 > $R/summary.json || exit 6
rm -rf $R/pmc1 $R/pmc2 $R/pmc3
ls -la $R
