#!/usr/bin/env python3
"""Audit of the kernels that are NOT this library's but run on its product paths (torch's own: noise, fills, casts, index gathers,
cat, reductions; RCCL's collective kernel) for the gfx950 packed-fp32 operand-selection erratum (docs/GFX950_PACKED_FP32_ERRATUM.md;
VERDICT r5 item 5).  Runs on the CPU container: the libraries are the ones of the image the GPU box runs.

    python tools/torch_kernel_audit.py --stats gpurun_out/r6d/*_kernel_stats.csv [--work /tmp/audit] [--out profiles/r6_torch_kernel_audit.txt]

1. kernel names per path from rocprofv3 `kernel_stats.csv` files (tools/paths_for_trace.py), library kernels (k_*) dropped;
2. the gfx950 code objects of libtorch_hip.so / librccl.so: `.hip_fatbin` is a sequence of COMPRESSED clang offload bundles
   ("CCOB", zstd); each is cut out by its header and unbundled for hipv4-amdgcn-amd-amdhsa--gfx950 with clang-offload-bundler;
3. symbols (llvm-readelf -s, demangled with c++filt) matched against the traced names; the matched functions disassembled
   (llvm-objdump --disassemble-symbols) and scanned with tools/isa_pk_opsel_scan.py's patterns: packed-fp32 instructions of any
   form, and the forms the isolated tests proved vulnerable (v_pk_{mul,add,fma}_f32 whose LOW result takes SRC1's HIGH half,
   op_sel:[0,1..], with SRC1 in VGPRs).
"""
import argparse
import csv
import glob
import os
import re
import struct
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("LAE_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
PK_ANY = re.compile(r"\bv_pk_(?:mul_f32|add_f32|fma_f32|mov_b32)\b")
PK_VULN = re.compile(r"\b(v_pk_(?:mul|add|fma)_f32)\s+(v\[\d+:\d+\])\s*,\s*([vs]\[\d+:\d+\]|[^,\s]+)\s*,\s*([vs]\[\d+:\d+\]|[^,\s]+)(.*)$")
OPSEL = re.compile(r"op_sel:\[(\d),(\d)")


def norm(name):
    """rocprofv3 prints demangled names with a leading 'void '; c++filt without: compare on the text without blanks"""
    n = name.strip()
    if n.startswith("void "):
        n = n[5:]
    return re.sub(r"\s+", "", n)


def library_kernel(name):
    n = name.replace("(anonymous namespace)::", "")
    return bool(re.match(r"^(void )?k_[a-z0-9_]+", n)) or "_GLOBAL__N_1" in name and re.search(r"\d+k_[a-z0-9_]+", name) is not None


def traced_names(stats_files):
    by_path = {}
    for f in stats_files:
        path = os.path.basename(f).replace("_kernel_stats.csv", "")
        for r in csv.DictReader(open(f)):
            if not library_kernel(r["Name"]):
                by_path.setdefault(r["Name"], {})[path] = int(r["Calls"])
    return by_path


def code_objects(so_path, work):
    """unbundle every gfx950 code object of a HIP shared library into work/<lib>/NNN.co (cached)"""
    tag = os.path.basename(so_path).split(".")[0]
    d = os.path.join(work, tag)
    if os.path.isdir(d) and glob.glob(d + "/*.co"):
        return sorted(glob.glob(d + "/*.co"))
    os.makedirs(d, exist_ok=True)
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so_path, os.path.join(d, "copy.so")])
    os.remove(os.path.join(d, "copy.so"))
    blob = open(fat, "rb").read()
    os.remove(fat)
    out, p, k = [], 0, 0
    while True:
        a = blob.find(b"CCOB", p)
        b = blob.find(b"__CLANG_OFFLOAD_BUNDLE__", p)
        cand = [x for x in (a, b) if x >= 0]
        if not cand:
            break
        p = min(cand)
        if p == a:
            ver = struct.unpack_from("<H", blob, p + 4)[0]
            total = struct.unpack_from("<I", blob, p + 8)[0] if ver == 2 else struct.unpack_from("<Q", blob, p + 8)[0]
            piece = blob[p:p + total]
            p += max(total, 4)
        else:                                                 # an uncompressed bundle: up to the next bundle magic
            nxt = [x for x in (blob.find(b"CCOB", p + 24), blob.find(b"__CLANG_OFFLOAD_BUNDLE__", p + 24)) if x >= 0]
            end = min(nxt) if nxt else len(blob)
            piece = blob[p:end]
            p = end
        bf = os.path.join(d, "bundle.bin")
        open(bf, "wb").write(piece)
        co = os.path.join(d, f"{k:04d}.co")
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "-type=o", "-targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"-input={bf}",
                            f"-output={co}", "-unbundle"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
            k += 1
        elif os.path.exists(co):
            os.remove(co)
    if os.path.exists(os.path.join(d, "bundle.bin")):
        os.remove(os.path.join(d, "bundle.bin"))
    return out


def functions_of(co):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", co], capture_output=True, text=True).stdout
    syms = sorted({ln.split()[-1] for ln in txt.splitlines() if " FUNC " in ln and len(ln.split()) >= 8})
    if not syms:
        return {}
    dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(syms, dem))


def scan_function(co, sym):
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f"--disassemble-symbols={sym}", co],
                         capture_output=True, text=True).stdout
    n_ins = n_pk = 0
    vuln = []
    for line in txt.splitlines():
        code = re.sub(r"//.*$", "", line)
        if re.match(r"^\s+[a-z_0-9]+\s", code):
            n_ins += 1
        if PK_ANY.search(code):
            n_pk += 1
        m = PK_VULN.search(code)
        if m:
            mn, dst, s0, s1, mods = m.groups()
            o = OPSEL.search(mods)
            if o and (o.group(1), o.group(2)) == ("0", "1") and s1.startswith("v[") and s1 != s0:
                vuln.append(code.strip())
    return n_ins, n_pk, vuln


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", nargs="+", required=True)
    ap.add_argument("--work", default="/tmp/lae_audit")
    ap.add_argument("--out", default=None)
    ap.add_argument("--libs", nargs="*", default=None)
    a = ap.parse_args()
    import torch
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    libs = a.libs or [os.path.join(libdir, "libtorch_hip.so"), os.path.join(libdir, "librccl.so")]
    names = traced_names(a.stats)
    want = {norm(n): n for n in names}
    found = {}
    for lib in libs:
        cos = code_objects(lib, a.work)
        print(f"{os.path.basename(lib)}: {len(cos)} gfx950 code objects", file=sys.stderr)
        for co in cos:
            for sym, dem in functions_of(co).items():
                key = norm(dem)
                if key in want and key not in found:
                    found[key] = (os.path.basename(lib), co, sym) + scan_function(co, sym)
    lines = []
    w = lines.append
    w("# kernels on the product paths that are not this library's, scanned for the gfx950 packed-fp32 erratum forms")
    w("# (tools/torch_kernel_audit.py; traces: tools/paths_for_trace.py under rocprofv3 --kernel-trace --stats)")
    w("# columns: instructions | packed-fp32 (any form) | vulnerable form (op_sel:[0,1], VGPR SRC1) | library | paths (calls) | kernel")
    tot_v = tot_pk = 0
    missing = []
    for key, full in sorted(want.items(), key=lambda kv: kv[1]):
        paths = ", ".join(f"{p} x{c}" for p, c in sorted(names[full].items()))
        short = re.sub(r"at::native::|\(anonymous namespace\)::|c10::|std::", "", full)[:150]
        if key not in found:
            missing.append((short, paths))
            continue
        lib, co, sym, n_ins, n_pk, vuln = found[key]
        tot_v += len(vuln); tot_pk += n_pk
        w(f"{n_ins:6d} | {n_pk:4d} | {len(vuln):3d} | {lib:16s} | {paths} | {short}")
        for v in vuln[:4]:
            w(f"           vulnerable: {v}")
    w(f"# {len(found)} kernels located and disassembled, {tot_pk} packed-fp32 instructions in them, {tot_v} of a vulnerable form")
    for short, paths in missing:
        w(f"# not located in the libraries' gfx950 code objects (runtime blit kernels of libamdhip64 / JIT): {short} [{paths}]")
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
