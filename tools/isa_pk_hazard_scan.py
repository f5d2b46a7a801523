#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for a packed-FP32 VALU instruction (v_pk_mul_f32 / v_pk_add_f32 /
v_pk_fma_f32) whose 64-bit result is read by the VERY NEXT instruction of the wave with no s_nop between them.

Why: round 4's "loop form" of k_grid_fwd_lean rendered frames that differed run to run beside a second process (DESIGN.md
section 8).  Its ISA differs from the passing form in one place that matters: LLVM's hazard recogniser puts `s_nop 0` behind a
VOP3P instruction only when op_sel_hi[0] is set (it reads that bit as VOP3's DST_OP_SEL: GCNHazardRecognizer,
getDstSelForwardingOperand), so `v_pk_mul_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[0,1]` followed by a reader of vD gets none.

    tools/isa_pk_hazard_scan.py file.s [kernel-name-substring]
prints every such pair with its kernel and line."""
import re
import sys

PK = re.compile(r"^\s*(v_pk_(?:mul|add|fma)_f32)\s+v\[(\d+):(\d+)\]\s*,\s*(.*)$")
REG_RANGE = re.compile(r"v\[(\d+):(\d+)\]")
REG = re.compile(r"\bv(\d+)\b")


def reads(ins, lo, hi):
    """does instruction text `ins` read any VGPR in [lo, hi]?  (every vector register operand after the first = a source;
    for stores / compares the first operand is a source too: treated as read as well -- conservative)"""
    body = ins.split(";")[0]
    ops = body.split(None, 1)
    if len(ops) < 2:
        return False
    mnem, rest = ops
    parts = rest.split(",")
    srcs = ",".join(parts[1:]) if not mnem.startswith(("global_store", "buffer_store", "ds_write", "ds_store", "flat_store", "v_cmp", "scratch_store")) else rest
    for m in REG_RANGE.finditer(srcs):
        a, b = int(m.group(1)), int(m.group(2))
        if a <= hi and b >= lo:
            return True
    for m in REG.finditer(REG_RANGE.sub("", srcs)):
        if lo <= int(m.group(1)) <= hi:
            return True
    return False


def main():
    path = sys.argv[1]
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    kernel, found, n_pk = None, [], 0
    lines = open(path).read().splitlines()
    code = []
    for i, ln in enumerate(lines):
        s = ln.split(";")[0].rstrip()
        m = re.match(r"^(_Z\w+|\w+):\s*$", s)
        if m and not s.startswith(".L"):
            kernel = m.group(1)
            continue
        if not s.strip() or s.lstrip().startswith(".") or s.rstrip().endswith(":"):
            continue
        code.append((i + 1, kernel, s.strip()))
    for k in range(len(code) - 1):
        ln, kern, ins = code[k]
        if only and only not in (kern or ""):
            continue
        m = PK.match(ins)
        if not m:
            continue
        n_pk += 1
        lo, hi = int(m.group(2)), int(m.group(3))
        nxt = code[k + 1][2]
        if nxt.startswith("s_nop"):
            continue
        if reads(nxt, lo, hi):
            found.append((kern, ln, ins, nxt))
    for kern, ln, ins, nxt in found:
        print(f"{kern}:{ln}\n    {ins}\n    {nxt}")
    print(f"{path}: {n_pk} packed-fp32 instructions, {len(found)} read back by the next instruction without s_nop")


if __name__ == "__main__":
    main()
