#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs per kernel: python3 scripts/pmc_summary.py <dir> [<dir> ...] > profiles/...json
Each <dir> holds one pass (*_counter_collection.csv).  Output: per kernel name, launches and the per-launch mean of
every counter found; FETCH_SIZE is reported raw (KB) and x2-corrected in bytes as MI355X_MICROARCH.md prescribes."""
import csv, glob, hashlib, json, os, sys, collections


def kernel_source_sha16():
    """Same hash as bench.py::kernel_source_sha16: ties the counters to the kernel sources they were measured with."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nir-gan_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(root)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(root, fn), "rb").read())
    return h.hexdigest()[:16]


acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
            cn, cv = r.get("Counter_Name"), r.get("Counter_Value")
            if not cn:
                continue
            a = acc[name][cn]
            a[0] += float(cv)
            a[1] += 1
out = {}
for name, cs in acc.items():
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
    short = short.split("(")[0].strip()
    e = {"launches": max(v[1] for v in cs.values())}
    for cn, (s, n) in cs.items():
        e[cn + "_per_launch"] = s / n
    if "FETCH_SIZE" in cs:
        e["hbm_read_bytes_per_launch_corrected"] = 2.0 * 1024.0 * cs["FETCH_SIZE"][0] / cs["FETCH_SIZE"][1]
    if "WRITE_SIZE" in cs:
        e["hbm_write_bytes_per_launch"] = 1024.0 * cs["WRITE_SIZE"][0] / cs["WRITE_SIZE"][1]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
        # MFMA-busy cycles are summed over the 4 SIMDs of 256 CUs; GRBM_GUI_ACTIVE over the 8 XCDs
        busy = cs["SQ_VALU_MFMA_BUSY_CYCLES"][0] / cs["SQ_VALU_MFMA_BUSY_CYCLES"][1]
        act = cs["GRBM_GUI_ACTIVE"][0] / cs["GRBM_GUI_ACTIVE"][1]
        e["mfma_busy_fraction_of_active_cycles"] = busy / (1024.0 * act / 8.0)
    out[short] = e
# whole run: every kernel of the command (5 steps of the train step: 1 warm-up + 3 timed + the loss read-out, plus the one-off engine build)
tb = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"][0] for cs in acc.values() if "SQ_VALU_MFMA_BUSY_CYCLES" in cs)
ta = sum(cs["GRBM_GUI_ACTIVE"][0] for cs in acc.values() if "GRBM_GUI_ACTIVE" in cs)
rd = sum(2.0 * 1024.0 * cs["FETCH_SIZE"][0] for cs in acc.values() if "FETCH_SIZE" in cs)
wr = sum(1024.0 * cs["WRITE_SIZE"][0] for cs in acc.values() if "WRITE_SIZE" in cs)
STEPS = int(os.environ.get("PMC_STEPS", "5"))
out["_whole_run"] = {"steps": STEPS,
                     "mfma_busy_fraction_of_active_cycles": (tb / (1024.0 * ta / 8.0)) if ta else None,
                     # v_mfma_f32_32x32x2_f32: 64 busy cycles per instruction per SIMD, 4096 FLOP each = 64 FLOP per busy cycle
                     "executed_mfma_gflop_per_step_fp32": tb * 64.0 / 1e9 / STEPS,
                     "hbm_read_gb_per_step": rd / 1e9 / STEPS, "hbm_write_gb_per_step": wr / 1e9 / STEPS,
                     "note": "sums over every kernel of the command; FETCH_SIZE x2-corrected (MI355X_MICROARCH.md); per step = / steps"}
out["_meta"] = {"kernel_src_sha16": kernel_source_sha16(), "command": "python3 bench.py --no-cpu-baseline --no-probe --sustain 0 --steps 3 --warmup 1",
                "passes": [os.path.basename(d.rstrip("/")) for d in sys.argv[1:]],
                "recorded_by": "scripts/refresh_profiles.sh (rocprofv3 --pmc, one counter group per pass)"}
json.dump(out, sys.stdout, indent=1, sort_keys=True)
