#!/bin/bash
# SQ counters (one pass) of the two F(4x4,3x3) plane-GEMM kernels on the microbenchmark: LDS 16-k stages (default), LDS 32-k stages.
# Round 2 also measured a register-fed variant (both operands fragment-major, one global_load_dwordx4 = one MFMA operand, no LDS, no
# barriers, four waves per SIMD): correct, but 81-90 TFLOP/s against 103-110 and MFMA busy 0.55-0.58 against 0.65-0.70 -- every wave loads
# its own fragments (2x the L1 traffic of the shared LDS image) and the vector-memory issue, not the barrier, is what starves the pipe; removed.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun}/gpurun_out/pmc_variants; rm -rf "$R"; mkdir -p "$R"
for s in gemm16 gemm32; do
  script=bench_wino6_gemm.py
  [ $s = gemm32 ] && export NIRGAN_WINO6_GEMM32=1
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$s -- python3 $GRAFT_REPO_ROOT/scripts/$script > $R/$s.log 2>&1 || echo fail $s
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_variants"
for s in ("gemm16","gemm32"):
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
    for fn in glob.glob(f"{R}/{s}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            n=r["Kernel_Name"]
            if "wino6" not in n: continue
            key=(n.split("(")[0][-24:], r["Grid_Size"])
            a=acc[key][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
    for k,cs in sorted(acc.items()):
        v={c:x[0]/x[1] for c,x in cs.items()}
        wc=v.get("SQ_WAVE_CYCLES",1)
        print(s,k,"launches",int(max(x[1] for x in cs.values())), " ".join(f"{c[3:]}={val/wc:.3f}" for c,val in v.items() if c.startswith("SQ_") and c!="SQ_WAVE_CYCLES"), "mfma_busy/active=%.3f"%(v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(1024*v.get("GRBM_GUI_ACTIVE",1)/8)))
PY
rm -rf $R/*/runc
