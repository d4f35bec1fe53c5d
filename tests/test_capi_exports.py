"""The C-ABI library loads and exports every symbol include/nirgan_hip.h declares (no compute calls: no GPU here)."""
import os
import re

from nirgan_hip import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "nirgan_hip.h")).read()
    declared = set(re.findall(r"\b(nirgan_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert declared == set(L.PROTOTYPES), declared ^ set(L.PROTOTYPES)
    L.check_exports()
    assert L.backend().nirgan_version() >= 100
    assert L.backend().nirgan_last_error() is not None


def test_descriptor_validation_rejects_bad_arguments_without_launching():
    """Argument errors come back as codes + message before any kernel launch (safe without a GPU)."""
    d = L.ConvDesc()
    rc = L.backend().nirgan_conv_igemm(d, None)
    assert rc == -1 and b"null" in L.backend().nirgan_last_error()
    w = L.WgradDesc()
    assert L.backend().nirgan_wgrad_igemm(w, None) == -1
    assert L.backend().nirgan_instnorm_ws_elems(2, 64, 64, 256) > 0
    be = L.backend()
    assert be.nirgan_image_metrics(L.MetricsDesc(), None) == -1 and b"image_metrics" in be.nirgan_last_error()
    assert be.nirgan_image_metrics_ws_elems(2, 256, 256) == 2 * 8 * 8 * 3
    assert be.nirgan_emd_loss(L.EmdLossDesc(), None) == -1 and b"emd_loss" in be.nirgan_last_error()
    assert be.nirgan_emd_loss_ws_bytes(3, 1000, 1) == 3 * 8 + 3 * 1000 * 4 and be.nirgan_emd_loss_ws_bytes(3, 1000, 0) == 24
    assert be.nirgan_ssim_loss(L.SsimLossDesc(), None) == -1 and b"ssim_loss" in be.nirgan_last_error()
    assert be.nirgan_ssim_loss_ws_elems(2, 64, 32, 11) == 3 * 2 * 64 * 32 + 3 * 2 * 74 * 42 + 2 * 2 * 1 and be.nirgan_ssim_loss_ws_elems(2, 64, 32, 4) == 0
    assert be.nirgan_location_encoder(L.LocEncDesc(), None) == -1 and b"location_encoder" in be.nirgan_last_error()
    assert be.nirgan_hist_match(L.HistMatchDesc(), None) == -1 and b"hist_match" in be.nirgan_last_error()
    assert be.nirgan_hist_match_ws_bytes(2, 65536) == 2 * 65536 * 12 and be.nirgan_hist_match_ws_bytes(1, 1000) == 2048 * 12
    assert be.nirgan_wino6_gemm(L.Wino6Desc(), None) == -1 and b"wino6_gemm" in be.nirgan_last_error()
    assert be.nirgan_wino6_input(L.Wino6Desc(), None) == -1 and be.nirgan_wino6_tiles(16, 64, 64) == 4096 and be.nirgan_wino6_tiles(2, 69, 66) == 2 * 18 * 17
    d = L.ConvDesc()
    d.precision = 7
    assert be.nirgan_conv_igemm(d, None) == -1


def test_struct_layouts_match_the_header(tmp_path):
    """ctypes mirrors have exactly the C layout: sizeof and the offset of the last field, computed by gcc."""
    import ctypes as C
    import subprocess
    pairs = [("nirgan_conv_desc", L.ConvDesc, "out_span"), ("nirgan_wgrad_desc", L.WgradDesc, "algo"),
             ("nirgan_in_fwd_desc", L.InFwdDesc, "y_bf16"), ("nirgan_in_bwd_desc", L.InBwdDesc, "g_bf16"),
             ("nirgan_tap_gather_desc", L.TapGatherDesc, "dst"), ("nirgan_tap_scatter_desc", L.TapScatterDesc, "dbias"),
             ("nirgan_pix_loss_desc", L.PixLossDesc, "ws_elems"), ("nirgan_inject_fwd_desc", L.InjectFwdDesc, "o_pad"),
             ("nirgan_inject_bwd_desc", L.InjectBwdDesc, "ws_elems"), ("nirgan_plan_entry", L.PlanEntry, "desc"),
             ("nirgan_metrics_desc", L.MetricsDesc, "means"), ("nirgan_locenc_desc", L.LocEncDesc, "features"),
             ("nirgan_hist_match_desc", L.HistMatchDesc, "out"), ("nirgan_ssim_loss_desc", L.SsimLossDesc, "grad_pred"), ("nirgan_emd_loss_desc", L.EmdLossDesc, "grad_pred"),
             ("nirgan_wino_dy_desc", L.WinoDyDesc, "r"), ("nirgan_wino6_desc", L.Wino6Desc, "U3")]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "nirgan_hip.h"\nint main(void){\n'
    for cname, _, last in pairs:
        src += f'printf("%zu %zu\\n", sizeof({cname}), offsetof({cname}, {last}));\n'
    src += "return 0;}\n"
    c = tmp_path / "layout.c"
    c.write_text(src)
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    for (cname, ct, last), line in zip(pairs, lines):
        size, off = (int(x) for x in line.split())
        assert C.sizeof(ct) == size, cname
        assert getattr(ct, ct._fields_[-1][0]).offset == off, cname


def test_integration_stub_mirrors_the_descriptor():
    """INTEGRATION.md shows the ctypes mirror a maintainer would write for nirgan_conv_desc: it has to match the library's own mirror
    (field names, order, size) -- the library reads the whole descriptor."""
    import ctypes as C
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    m = re.search(r"class ConvDesc\(C\.Structure\):.*?\n((?:    .*\n)+)", text)
    assert m, "INTEGRATION.md: the ConvDesc stub is gone"
    ns = {"C": C}
    exec("class ConvDesc(C.Structure):\n" + m.group(1), ns)
    stub = ns["ConvDesc"]
    assert [f[0] for f in stub._fields_] == [f[0] for f in L.ConvDesc._fields_]
    assert C.sizeof(stub) == C.sizeof(L.ConvDesc)
