"""gfx950 packed-fp32 operand-selection erratum: while v_mfma instructions of another wave are in
flight on a SIMD, `v_pk_mul_f32 / v_pk_add_f32 vD, SRC0, SRC1 op_sel:[0,1]` with SRC1 a VGPR pair other than SRC0 sometimes
computes its low result with SRC1's high half read as zero (tools/ubench/pk_opsel.hip, profiles/r5_pk_opsel_erratum.txt).  The
build swaps the operands of every such instruction (laenerf_amd/build.py); these tests pin the rewrite and scan the SHIPPED code
object for survivors with an independent pattern (tools/isa_pk_opsel_scan.py)."""
import json
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_rewrite_swaps_sources_with_their_modifiers_and_leaves_safe_forms_alone():
    from laenerf_amd import build as B
    cases = {
        "\tv_pk_mul_f32 v[20:21], v[6:7], v[22:23] op_sel:[0,1] op_sel_hi:[0,1]":
            "\tv_pk_mul_f32 v[20:21], v[22:23], v[6:7] op_sel:[1,0] op_sel_hi:[1,0]",
        "\tv_pk_add_f32 v[24:25], v[2:3], v[8:9] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]":
            "\tv_pk_add_f32 v[24:25], v[8:9], v[2:3] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]",
        "\tv_pk_mul_f32 v[62:63], v[98:99], v[90:91] op_sel:[0,1]":                      # default op_sel_hi = [1,1]
            "\tv_pk_mul_f32 v[62:63], v[90:91], v[98:99] op_sel:[1,0] op_sel_hi:[1,1]",
    }
    for before, after in cases.items():
        got, n = B.pk_erratum_rewrite(before)
        assert n == 1 and got == after, (before, got)
    untouched = ["\tv_pk_mul_f32 v[0:1], v[0:1], s[12:13] op_sel:[0,1]",                 # SGPR source: never wrong
                 "\tv_pk_add_f32 v[26:27], v[26:27], v[26:27] op_sel:[0,1] op_sel_hi:[1,0]",   # SRC1 == SRC0: never wrong
                 "\tv_pk_mul_f32 v[20:21], v[22:23], v[6:7] op_sel:[1,0] op_sel_hi:[1,0]",
                 "\tv_pk_fma_f32 v[4:5], v[4:5], v[8:9], v[6:7] op_sel:[0,1,0] op_sel_hi:[0,1,1]",
                 "\tv_pk_mov_b32 v[4:5], v[4:5], v[8:9] op_sel:[0,1]",
                 "\tv_pk_mul_f16 v4, v5, v8 op_sel:[0,1]"]
    for line in untouched:
        got, n = B.pk_erratum_rewrite(line)
        assert n == 0 and got == line, line


def test_scanner_pattern_agrees_with_the_build_pattern():
    import isa_pk_opsel_scan as S
    from laenerf_amd import build as B
    lines = ["v_pk_mul_f32 v[20:21], v[6:7], v[22:23] op_sel:[0,1] op_sel_hi:[0,1]", "v_pk_add_f32 v[8:9], v[2:3], v[8:9] op_sel:[0,1] op_sel_hi:[1,0]",
             "v_pk_mul_f32 v[0:1], v[0:1], s[12:13] op_sel:[0,1]", "v_pk_add_f32 v[26:27], v[26:27], v[26:27] op_sel:[0,1] op_sel_hi:[1,0]",
             "v_pk_mul_f32 v[20:21], v[22:23], v[6:7] op_sel:[1,0] op_sel_hi:[1,0]", "v_pk_fma_f32 v[4:5], v[4:5], s[0:1], v[6:7]"]
    for ln in lines:
        assert (S.vulnerable(ln) is not None) == (B.pk_erratum_is_vulnerable("\t" + ln) is not None), ln
    assert S.vulnerable("v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] clamp") is not None      # the scanner is the looser of the two


def test_shipped_code_object_has_no_vulnerable_packed_fp32_instruction(hip_lib):
    import isa_pk_opsel_scan as S
    from laenerf_amd import _lib
    texts = S.disassemble_so(_lib.SO_PATH)
    assert len(texts) >= 8                                        # one code object per .hip file
    n_pk = hits = 0
    names = set()
    for t in texts:
        h, n = S.scan_text(t)
        hits += len(h); n_pk += n
        names.update(ln.split("<")[1].split(">")[0] for ln in t.splitlines() if ">:" in ln and "<" in ln)
    assert n_pk > 5000 and any("k_grid_fwd_lean" in k for k in names) and any("k_bwd_walk" in k for k in names)   # the disassembly is the library's
    assert hits == 0
    meta = json.load(open(_lib.SO_PATH + ".isa.json"))
    assert meta["pk_erratum_rewrite"] is True and meta["total"] >= 1 and meta["instructions_rewritten"]["gridencoder.hip"] >= 1
