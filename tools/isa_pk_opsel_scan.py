#!/usr/bin/env python3
"""Scan gfx950 assembly / disassembly for the packed-fp32 form that returns wrong results while the matrix pipe is busy
(tools/ubench/pk_opsel.hip; profiles/r5_pk_opsel_erratum.txt):

    v_pk_mul_f32 / v_pk_add_f32  vD, SRC0, SRC1  op_sel:[0,1] ...     with SRC1 a VGPR pair other than SRC0

(the LOW result takes SRC1's high half, which is sometimes read as zero while v_mfma instructions of another wave are in flight
on the SIMD).  SGPR SRC1, SRC1 == SRC0, v_pk_fma_f32, v_pk_mov_b32 and every other op_sel value were never wrong in 1e9 trials each.

    tools/isa_pk_opsel_scan.py file.s [...]        -> lists every such instruction with its kernel; exit code 1 if any
    tools/isa_pk_opsel_scan.py --so lib.so         -> the same over the gfx950 code objects embedded in a HIP shared library
                                                      (.hip_fatbin bundles, disassembled with llvm-objdump)
v_pk_fma_f32 with op_sel:[0,1,..] turned out to be vulnerable too (k_grid_fwd's dy_dx chain, lanes 48-63 of every wave beside an
MFMA neighbour), so the shipped library is built WITHOUT packed-fp32 instructions (laenerf_amd/build.py); `--so` also reports how
many packed-fp32 instructions of any form a library holds, and tests/test_isa_cpu.py asserts that number is 0 for the shipped one."""
import re
import sys

INS = re.compile(r"\b(v_pk_(?:mul|add)_f32)\s+(v\[\d+:\d+\])\s*,\s*([vs]\[\d+:\d+\]|[^,\s]+)\s*,\s*([vs]\[\d+:\d+\]|[^,\s]+)(.*)$")
OPSEL = re.compile(r"op_sel:\[(\d),(\d)\]")


def vulnerable(line):
    """(mnemonic, dst, src0, src1, modifiers) if `line` is the vulnerable form, else None"""
    m = INS.search(line.split(";")[0])
    if not m:
        return None
    mn, dst, s0, s1, mods = m.groups()
    o = OPSEL.search(mods)
    if not o or (o.group(1), o.group(2)) != ("0", "1"):
        return None
    if not s1.startswith("v[") or s1 == s0:
        return None
    return mn, dst, s0, s1, mods.strip()


def scan(path):
    kernel, hits = None, []
    for n, line in enumerate(open(path, errors="replace"), 1):
        k = re.match(r"^(?:[0-9a-f]+\s+<)?(_Z\w+|[A-Za-z_]\w*)>?:\s*(?:;.*)?$", line.strip())
        if k and not line.lstrip().startswith("."):
            kernel = k.group(1)
        if vulnerable(line):
            hits.append((kernel, n, line.strip()))
    return hits


def code_objects_of(so_path):
    """the gfx950 ELF images inside a HIP shared library: .hip_fatbin is a sequence of clang offload bundles
    (magic, u64 entries, per entry u64 offset / u64 size / u64 triple length / triple)"""
    import struct
    import subprocess
    import tempfile
    objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
    with tempfile.TemporaryDirectory() as td:
        fat = f"{td}/fat.bin"
        subprocess.check_call([objcopy, "--dump-section", f".hip_fatbin={fat}", so_path, f"{td}/copy.so"])
        blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], 0
    while True:
        pos = blob.find(magic, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", blob, pos + 24)[0]
        q = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos += 24
    return out


def disassemble_so(so_path):
    import subprocess
    import tempfile
    texts = []
    for k, img in enumerate(code_objects_of(so_path)):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img); f.flush()
            texts.append(subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", f.name], stdout=subprocess.PIPE, check=True).stdout.decode(errors="replace"))
    return texts


PK_FP32 = re.compile(r"\bv_pk_(?:mul_f32|add_f32|fma_f32|mov_b32)\b")


def scan_text(text, counts=None):
    """-> (instructions of the form the isolated test proved vulnerable, packed instructions of any kind); counts (dict, optional)
    receives the number of packed-FP32 instructions of any form under the key "packed_fp32"."""
    kernel, hits, n_pk = None, [], 0
    for line in text.splitlines():
        k = re.match(r"^[0-9a-f]+\s+<([^>]+)>:", line.strip())
        if k:
            kernel = k.group(1)
        code = re.sub(r"//.*$", "", line)
        if "v_pk_" in code:
            n_pk += 1
            if counts is not None and PK_FP32.search(code):
                counts["packed_fp32"] = counts.get("packed_fp32", 0) + 1
        if vulnerable(code):
            hits.append((kernel, line.strip()))
    return hits, n_pk


def main():
    total = 0
    if len(sys.argv) > 2 and sys.argv[1] == "--so":
        n_pk = 0
        counts = {}
        texts = disassemble_so(sys.argv[2])
        for t in texts:
            hits, n = scan_text(t, counts)
            n_pk += n
            total += len(hits)
            for kernel, line in hits:
                print(f"{kernel}: {line}")
        print(f"{sys.argv[2]}: {len(texts)} code object(s), {n_pk} packed instructions, {counts.get('packed_fp32', 0)} of them packed-fp32 "
              f"(any form), {total} of the form the isolated test proved vulnerable")
        sys.exit(1 if counts.get("packed_fp32", 0) else 0)
    for path in sys.argv[1:]:
        hits = scan(path)
        total += len(hits)
        for kernel, n, line in hits:
            print(f"{path}:{n}: {kernel}: {line}")
    print(f"{total} vulnerable packed-fp32 instruction(s)")
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
