"""gfx950 packed-fp32 operand-selection erratum: while v_mfma instructions of another wave are in
flight on a SIMD, a packed-fp32 instruction whose LOW result takes SRC1's HIGH half can compute it with SRC1.hi read as zero
(v_pk_mul_f32 / v_pk_add_f32: tools/ubench/pk_opsel.hip, profiles/r5_pk_opsel_erratum.txt; v_pk_fma_f32: k_grid_fwd's dy_dx chain,
profiles/r5_suite_beside_mfma.txt).  The compiler emits such forms for ordinary HIP code, and which forms are safe is only known
empirically, so the library is built WITHOUT packed-fp32 instructions (laenerf_amd/build.py).  These tests pin the build's check
and scan the SHIPPED code object with an independent pattern (tools/isa_pk_opsel_scan.py)."""
import json

import pytest
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_build_check_recognises_packed_fp32_instructions_and_nothing_else():
    from laenerf_amd import build as B
    asm = "\n".join(["\tv_pk_mul_f32 v[20:21], v[6:7], v[22:23] op_sel:[0,1] op_sel_hi:[0,1]", "\tv_pk_add_f32 v[8:9], v[2:3], v[8:9] neg_lo:[0,1] neg_hi:[0,1]",
                     "\tv_pk_fma_f32 v[6:7], v[14:15], v[6:7], v[12:13] op_sel:[0,1,0]   ; comment", "\tv_pk_mov_b32 v[4:5], v[4:5], v[8:9] op_sel:[0,1]",
                     "\tv_pk_mul_f16 v4, v5, v8 op_sel:[0,1]", "\tv_pk_add_f16 v5, v5, v6", "\tv_cvt_pk_f16_f32 v6, v18, v19", "\tv_mul_f32_e32 v6, v20, v2",
                     "\t; v_pk_mul_f32 in a comment", "\tv_pk_fma_f16 v1, v2, v3, v4"])
    found = B.packed_fp32_instructions(asm)
    assert len(found) == 4 and all(f.startswith(("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mov_b32")) for f in found)
    assert B.NO_PACKED_FP32 == ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def test_scanner_flags_the_form_the_isolated_test_proved_vulnerable():
    import isa_pk_opsel_scan as S
    yes = ["v_pk_mul_f32 v[20:21], v[6:7], v[22:23] op_sel:[0,1] op_sel_hi:[0,1]", "v_pk_add_f32 v[8:9], v[2:3], v[8:9] op_sel:[0,1] op_sel_hi:[1,0]",
           "v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] clamp"]
    no = ["v_pk_mul_f32 v[0:1], v[0:1], s[12:13] op_sel:[0,1]", "v_pk_add_f32 v[26:27], v[26:27], v[26:27] op_sel:[0,1] op_sel_hi:[1,0]",
          "v_pk_mul_f32 v[20:21], v[22:23], v[6:7] op_sel:[1,0] op_sel_hi:[1,0]", "v_pk_mul_f16 v4, v5, v8 op_sel:[0,1]"]
    assert all(S.vulnerable(ln) is not None for ln in yes) and all(S.vulnerable(ln) is None for ln in no)
    counts = {}
    hits, n_pk = S.scan_text("0000 <k>:\n\t" + "\n\t".join(yes + no + ["v_pk_fma_f32 v[6:7], v[14:15], v[6:7], v[12:13] op_sel:[0,1,0]"]), counts)
    assert len(hits) == 3 and n_pk == 8 and counts["packed_fp32"] == 7            # every packed-FP32 form counts, safe-looking or not


def test_shipped_code_object_has_no_packed_fp32_instruction(hip_lib):
    import isa_pk_opsel_scan as S
    from laenerf_amd import _lib
    texts = S.disassemble_so(_lib.SO_PATH)
    assert len(texts) >= 8                                        # one code object per .hip file
    counts, n_pk, names = {}, 0, set()
    for t in texts:
        h, n = S.scan_text(t, counts)
        n_pk += n
        names.update(ln.split("<")[1].split(">")[0] for ln in t.splitlines() if ">:" in ln and "<" in ln)
    # the disassembly is the library's: its kernels are there, and so are the 16-bit packed forms (never wrong in 1e9 trials each)
    assert n_pk > 2000 and any("k_grid_fwd_lean" in k for k in names) and any("k_bwd_walk" in k for k in names)
    assert counts.get("packed_fp32", 0) == 0
    meta = json.load(open(_lib.SO_PATH + ".isa.json"))
    assert meta["packed_fp32_ops"] is False and meta["total"] == 0


def test_packed_fp32_build_never_targets_the_default_library(monkeypatch):
    """packed_fp32 / LAE_BUILD_PACKED_FP32=1 are probe-build switches: refused without out=<another file> and for
    out=<the default library>; nothing is compiled before the refusal"""
    from laenerf_amd import build
    called = []
    monkeypatch.setattr(build, "_build", lambda *a, **k: called.append(a))
    with pytest.raises(RuntimeError):
        build.build(packed_fp32=True)
    with pytest.raises(RuntimeError):
        build.build(out=build.SO, packed_fp32=True)
    monkeypatch.setenv("LAE_BUILD_PACKED_FP32", "1")
    with pytest.raises(RuntimeError):
        build.build(force=True)
    assert not called
    build.build(out="/tmp/lae_probe_never_built.so")          # with out= the variable is honoured: _build(verbose, flags, packed, target)
    assert len(called) == 1 and called[0][2] is True and called[0][3] == "/tmp/lae_probe_never_built.so"
    assert build.default_is_packed() is False                 # the shipped library's record
