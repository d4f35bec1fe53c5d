"""scripts/check_x3_asm.py: the build check of the hand-ordered split tiles (no GPU: hipcc cross-compiles).  The rule is exercised on
synthetic instruction streams first (it must FIND what it is there to find), then on the shipped kernels."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_x3_asm", os.path.join(ROOT, "scripts", "check_x3_asm.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def _loop(mid):
    asm = lambda s: (s, True)
    cc = lambda s: (s, False)
    return [cc(".LBB0_1:")] + mid + [cc("v_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], a[8:11], a[0:3]"), cc("s_cbranch_scc0 .LBB0_1")]


def test_rule_finds_a_copy_in_front_of_the_covering_wait():
    asm = lambda s: (s, True)
    cc = lambda s: (s, False)
    ok = _loop([asm("global_load_dwordx4 v[10:13], v1, s[2:3] offset:0"), asm("global_load_dwordx4 v[14:17], v1, s[2:3] offset:16"),
                asm("s_waitcnt vmcnt(1)"), cc("v_add_f32 v20, v10, v11"), asm("s_waitcnt vmcnt(0)"), cc("v_add_f32 v21, v14, v15")])
    bad, n = chk.check_async("t", ok)
    assert n == 2 and not bad, bad
    # a copy of the second load's destination between its issue and the wait that covers it (vmcnt(1) covers only the first load)
    copy = _loop([asm("global_load_dwordx4 v[10:13], v1, s[2:3] offset:0"), asm("global_load_dwordx4 v[14:17], v1, s[2:3] offset:16"),
                  asm("s_waitcnt vmcnt(1)"), cc("v_mov_b32 v30, v14"), asm("s_waitcnt vmcnt(0)")])
    bad, _ = chk.check_async("t", copy)
    assert len(bad) == 1 and "v_mov_b32 v30, v14" in bad[0]
    # a spill right behind the load (what hipcc did under register pressure)
    spill = _loop([asm("global_load_dwordx4 v[10:13], v1, s[2:3] offset:0"), cc("scratch_store_dwordx4 off, v[10:13], off"), asm("s_waitcnt vmcnt(0)")])
    bad, _ = chk.check_async("t", spill)
    assert len(bad) == 1 and "scratch_store" in bad[0]
    # fragment reads: the wait must count the LDS operations issued behind the read
    frag = _loop([asm("ds_read_b128 a[0:3], v5 offset:0"), asm("ds_read_b128 a[4:7], v5 offset:1024"), asm("s_waitcnt lgkmcnt(1)"),
                  cc("v_mfma_f32_16x16x32_bf16 a[16:19], v[0:3], a[4:7], a[16:19]"), asm("s_waitcnt lgkmcnt(0)")])
    bad, _ = chk.check_async("t", frag)
    assert len(bad) == 1 and "a[4:7]" in bad[0]


def test_rule_finds_a_write_to_store_data_within_two_wait_states():
    asm = lambda s: (s, True)
    cc = lambda s: (s, False)
    body = [asm("global_store_dwordx4 v9, v[102:105], s[40:41]"), asm("s_nop 0"), cc("v_pk_add_f32 v[104:105], v[96:97], v[124:125]")]
    bad, n = chk.check_stores("t", body)
    assert n == 1 and len(bad) == 1
    body[1] = asm("s_nop 1")
    assert not chk.check_stores("t", body)[0]


def test_shipped_kernels_pass_the_check():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_x3_asm.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "conv_x3r_kernel<128, plain>" in r.stdout and "vgpr_spill_count 0" in r.stdout
