#!/usr/bin/env python3
"""Build check of the hand-ordered split tiles (csrc/igemm_x3.h, igemm_x3r.h): compiles csrc/igemm_conv.hip to gfx950 assembly (hipcc
cross-compiles, no GPU) and checks, per kernel,

  1. register spills: .vgpr_spill_count / .private_segment_fixed_size of the kernel descriptors (the four-wave plain kernel must have
     none: a scratch reload inside its K loop is a `s_waitcnt vmcnt(0)` -- every fetch in flight drained);
  2. asynchronous destinations: the fragment reads (`ds_read_b128`) and the raw-row / weight-piece fetches (`global_load_dwordx4`) of
     these kernels are inline asm with hand-counted waits, so the compiler believes their destination registers valid at once.  Nothing
     may read or write such a register between the instruction and the wait that covers it (a copy inserted there copies stale data:
     seen twice while these kernels were written -- a register spilled right behind its load, and phi copies in front of a wait that
     stood in one arm of an `if`).  The covering wait of an operation is the first `s_waitcnt` of its counter whose count is not larger
     than the number of operations of that counter issued behind it (the counters retire in order);
  3. stores: a VALU instruction writing a data register of a `global_store_dwordx4` within two wait states of it (the store reads its
     data late on gfx950; the compiler's hazard recognizer does not see into inline asm).

Exit status 1 on a violation.   python scripts/check_x3_asm.py [--keep DIR]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "nir-gan_amd", "csrc", "igemm_conv.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
KERNELS = {                                     # substring of the mangled name -> (label, spills allowed)
    "conv_x3r_kernelILi128ELi0E": ("conv_x3r_kernel<128, plain>", False),
    "conv_x3r_kernelILi128ELi1E": ("conv_x3r_kernel<128, statistics>", False),
    "conv_x3r_kernelILi128ELi2E": ("conv_x3r_kernel<128, fused pass>", True),
    "conv_x3_kernelILi128": ("conv_x3_kernel<128>", True),
    "conv_x3_kernelILi64": ("conv_x3_kernel<64>", True),
}
VM_OPS = ("global_load", "global_store", "buffer_load", "buffer_store", "scratch_load", "scratch_store", "global_atomic")
LGKM_OPS = ("ds_read", "ds_write", "ds_bpermute", "ds_swizzle", "ds_permute")


def regs(tok):
    """register numbers named by an operand token: v12, v[3:6], a[0:3], a7 -> {('v', 12), ...}"""
    out = set()
    for kind, lo, hi in re.findall(r"\b([va])\[(\d+):(\d+)\]", tok):
        out.update((kind, i) for i in range(int(lo), int(hi) + 1))
    for kind, n in re.findall(r"\b([va])(\d+)\b", tok):
        out.add((kind, int(n)))
    return out


def parse(path):
    """{kernel: [(text, in_asm)]} for the kernels of KERNELS + {kernel: metadata dict}"""
    bodies, meta, cur, inasm = {}, {}, None, False
    name = None
    for line in open(path):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            cur = next((k for k in KERNELS if k in m.group(1)), None)
            if cur:
                bodies[cur] = []
            continue
        if s.startswith(".Lfunc_end"):
            cur = None
        if s.startswith(".name:"):
            name = next((k for k in KERNELS if k in s), None)
            if name:
                meta.setdefault(name, {})
        elif name and re.match(r"\.(vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|vgpr_count|agpr_count):", s):
            k, v = s.split(":")
            meta[name][k.strip(".")] = int(v)
        if cur is None:
            continue
        if s.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if s.startswith(";;#ASMEND"):
            inasm = False
            continue
        code = s.split(";")[0].strip()
        if code and not code.startswith("."):
            bodies[cur].append((code, inasm))
        elif code.endswith(":"):
            bodies[cur].append((code, inasm))
    return bodies, meta


def loop_span(body, whole=True):
    """index range [lo, hi] (label .. the backward branch to it) of the innermost loop that holds ALL the MFMAs -- the walk over a
    workgroup's items: first K-tile, K loop, epilogue -- or, whole=False, of the innermost loop that holds any: the K loop itself"""
    mf = [i for i, (c, _) in enumerate(body) if c.startswith("v_mfma")]
    if not mf:
        return None
    if not whole:
        labels = {c[:-1]: i for i, (c, _) in enumerate(body) if c.endswith(":")}
        best = None
        for i, (c, _) in enumerate(body):
            m = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", c)
            tgt = labels.get(m.group(1) or m.group(2)) if m else None
            if tgt is not None and tgt < i and any(tgt <= k <= i for k in mf) and (best is None or (i - tgt) < (best[1] - best[0])):
                best = (tgt, i)
        return best
    labels = {c[:-1]: i for i, (c, _) in enumerate(body) if c.endswith(":")}
    best = None
    for i, (c, _) in enumerate(body):
        m = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", c)
        if not m:
            continue
        tgt = labels.get(m.group(1) or m.group(2))
        if tgt is not None and tgt <= mf[0] and i >= mf[-1]:
            if best is None or (i - tgt) < (best[1] - best[0]):
                best = (tgt, i)
    return best


def check_async(label, body):
    """rule 2 over the K loop taken cyclically"""
    span = loop_span(body)
    if span is None:
        return [f"{label}: no loop around the MFMAs found"], 0
    lo, hi = span
    loop = body[lo:hi + 1]
    n = len(loop)
    bad, checked = [], 0
    for i, (code, inasm) in enumerate(loop):
        if not inasm:
            continue
        is_ld = code.startswith("global_load_dwordx4")
        is_ds = code.startswith("ds_read_b128")
        if not (is_ld or is_ds):
            continue
        dest = regs(code.split(",")[0])
        counter, ops = ("vmcnt", VM_OPS) if is_ld else ("lgkmcnt", LGKM_OPS)
        behind, covered = 0, False
        for step in range(1, 2 * n):
            c2, _ = loop[(i + step) % n]
            m = re.search(counter + r"\((\d+)\)", c2) if c2.startswith("s_waitcnt") else None
            if m and int(m.group(1)) <= behind:
                covered = True
                break
            if c2.startswith("s_waitcnt") or c2.endswith(":") or c2.startswith("s_nop"):
                continue
            if regs(c2) & dest:
                bad.append(f"{label}: `{c2}` touches the destination of `{code}` before the {counter} wait that covers it")
                break
            if c2.startswith(ops):
                behind += 1
        checked += 1
        if not covered and not bad:
            bad.append(f"{label}: no covering {counter} wait found for `{code}` inside the K loop")
    return bad, checked


def check_stores(label, body):
    """rule 3: two wait states behind a 16-byte store issued from inline asm"""
    bad, checked = [], 0
    for i, (code, inasm) in enumerate(body):
        if not (inasm and code.startswith("global_store_dwordx4")):
            continue
        data = regs(code.split(",")[1])
        checked += 1
        states = 0
        for c2, _ in body[i + 1:i + 4]:
            if c2.startswith("s_nop"):
                states += int(c2.split()[1]) + 1
                continue
            if states >= 2:
                break
            if c2.startswith("v_") and regs(c2.split(",")[0]) & data:
                bad.append(f"{label}: `{c2}` writes a data register of `{code}` {states} wait state(s) behind it")
            if not c2.startswith(("global_store", "s_")):
                states += 1
    return bad, checked


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    with tempfile.TemporaryDirectory() as tmp:
        out = keep or tmp
        os.makedirs(out, exist_ok=True)
        asm = os.path.join(out, "igemm_conv.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", "--cuda-device-only", "-S", SRC, "-o", asm], check=True)
        bodies, meta = parse(asm)
    problems = []
    for key, (label, spills_ok) in KERNELS.items():
        if key not in bodies:
            problems.append(f"{label}: kernel not found in the assembly")
            continue
        md = meta.get(key, {})
        line = (f"{label}: vgpr {md.get('vgpr_count')} (+agpr {md.get('agpr_count', 0)}), vgpr_spill_count {md.get('vgpr_spill_count')}, "
                f"sgpr_spill_count {md.get('sgpr_spill_count')}, private_segment_fixed_size {md.get('private_segment_fixed_size')}")
        if not spills_ok and (md.get("vgpr_spill_count", 1) != 0 or md.get("private_segment_fixed_size", 1) != 0):
            problems.append(label + ": spills to scratch")
        scratch_in_loop = 0
        span = loop_span(bodies[key], whole=False)
        if span:
            scratch_in_loop = sum(1 for c, _ in bodies[key][span[0]:span[1] + 1] if c.startswith("scratch_"))
            if scratch_in_loop and key.startswith("conv_x3r"):
                problems.append(f"{label}: {scratch_in_loop} scratch operations inside the K loop")
        b1, n1 = check_async(label, bodies[key])
        b2, n2 = check_stores(label, bodies[key])
        problems += b1 + b2
        print(f"{line}; scratch operations in the K loop {scratch_in_loop}; {n1} asynchronous destinations and {n2} asm stores checked"
              + (": ok" if not (b1 or b2) else ": VIOLATIONS"))
    for p in problems:
        print("PROBLEM:", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
