"""Build laenerf_amd/lib/liblaenerf_hip.so (HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU; the resulting .so is kept in-tree (git-ignored) so it
travels to the GPU box with the snapshot.  `python -m laenerf_amd.build [--force]`.

Device code goes through its ASSEMBLY (round 5): every .hip file is compiled to gfx950 assembly, `pk_erratum_rewrite` swaps the
operands of the one packed-fp32 instruction form that returns wrong results while the matrix pipe is busy (see below), the
assembly is assembled / linked / bundled with the LLVM tools of the ROCm image, and the host half of the file is compiled
against that code object (`-fcuda-include-gpubinary`) -- the same steps `hipcc -c` runs internally, with one pass in between.
"""
import concurrent.futures
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "liblaenerf_hip.so")
SOURCES = ["raymarching.hip", "gridencoder.hip", "shencoder.hip", "freqencoder.hip", "ffmlp.hip", "densitygrid.hip", "optimizer.hip", "loss.hip", "palette.hip", "editgrid.hip", "lae_common.cpp"]
# -ffp-contract=off: only explicit fmaf() fuses -> bit-identical sample indices/positions vs the oracle
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         # MFMA accumulators in VGPRs: with the default AGPR form every accumulator tile is copied back with
         # v_accvgpr_read before the VALU epilogue (88 copies per 16-row tile in the fused head, 1036 -> 940 instructions)
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
LLVM_BIN = os.environ.get("LAE_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

# ---------------------------------------------------------------------------------------------------------------------------
# gfx950 packed-fp32 operand-selection erratum (found in round 5; DESIGN.md section 8, tools/ubench/pk_opsel.hip,
# profiles/r5_pk_opsel_erratum.txt).  While v_mfma instructions of ANOTHER wave are in flight on the SIMD (a kernel on a second
# stream, or a second process on the GPU),
#       v_pk_mul_f32 / v_pk_add_f32  vD, SRC0, SRC1  op_sel:[0,1] ...      with SRC1 a VGPR pair other than SRC0
# sometimes computes its LOW result with SRC1's high half read as ZERO (3-10 % of the instructions beside back-to-back MFMA loops,
# 0 of 1e9 alone).  Every other op_sel value, an SGPR SRC1, SRC1 == SRC0, v_pk_fma_f32 and v_pk_mov_b32 were never wrong.  The
# compiler emits the form wherever it folds a lane shuffle into a packed multiply / add (55 places in this library: the fill pass
# of the hash-grid backward, the SH encoder, the palette backward, the frustum marking).  Both operations commute, so the same
# arithmetic is available as  vD, SRC1, SRC0  op_sel:[1,0]  with the per-source modifiers swapped -- bit-identical results, and
# the form that was never wrong.  tests/test_isa_cpu.py scans the shipped code object for survivors.
_PK = re.compile(r"^(\s*)(v_pk_(?:mul|add)_f32)(\s+)(v\[\d+:\d+\])\s*,\s*([vs]\[\d+:\d+\]|[^,\s]+)\s*,\s*([vs]\[\d+:\d+\]|[^,\s]+)((?:\s+\w+:\[[\d,]+\])*)\s*(;.*)?$")
_MOD = re.compile(r"(\w+):\[(\d),(\d)\]")


def pk_erratum_is_vulnerable(line):
    m = _PK.match(line.rstrip("\n"))
    if not m:
        return None
    mods = dict((k, (a, b)) for k, a, b in _MOD.findall(m.group(7) or ""))
    if mods.get("op_sel") != ("0", "1"):
        return None
    s0, s1 = m.group(5), m.group(6)
    if not s1.startswith("v[") or s1 == s0:
        return None
    return m, mods


def pk_erratum_rewrite(asm_text):
    """-> (rewritten assembly, number of instructions rewritten)"""
    out, n = [], 0
    for line in asm_text.split("\n"):
        hit = pk_erratum_is_vulnerable(line)
        if hit is None:
            out.append(line)
            continue
        m, mods = hit
        mods.setdefault("op_sel_hi", ("1", "1"))
        swapped = " ".join(f"{k}:[{b},{a}]" for k, (a, b) in mods.items())
        out.append(f"{m.group(1)}{m.group(2)}{m.group(3)}{m.group(4)}, {m.group(6)}, {m.group(5)} {swapped}")
        n += 1
    return "\n".join(out), n


def _deps():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    out.append(os.path.join(HERE, "..", "include", "laenerf.h"))
    out.append(os.path.abspath(__file__))
    return out


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False, out=None, extra_flags=(), rewrite=True):
    """out / extra_flags / rewrite=False: probe builds (tools/grid_loop_fault.sh): another library file, never the in-tree default"""
    global SO
    if out is not None:
        saved, SO = SO, os.path.abspath(out)
        try:
            os.makedirs(os.path.dirname(SO), exist_ok=True)
            return _build(verbose, list(extra_flags), rewrite)
        finally:
            SO = saved
    if not force and not needs_build():
        return SO
    return _build(verbose, [], rewrite)


def _run(cmd, verbose, what):
    if verbose:
        print(" ".join(cmd))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if p.returncode != 0:
        sys.stderr.write(p.stdout.decode())
        raise RuntimeError(f"{what} failed")
    if verbose and p.stdout:
        print(p.stdout.decode())


def _compile_one(src, obj, flags, verbose, rewrite):
    """one translation unit: device assembly -> erratum rewrite -> code object -> fat binary -> host object.  -> instructions rewritten"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    path = os.path.join(CSRC, src)
    stem = obj[:-2]
    asm, dev_o, co, fb = stem + ".s", stem + ".dev.o", stem + ".co", stem + ".hipfb"
    try:
        _run([hipcc] + flags + ["-x", "hip", "--cuda-device-only", "-S", path, "-o", asm], verbose, f"hipcc (device) on {src}")
        text = open(asm).read()
        n = 0
        if rewrite:
            text, n = pk_erratum_rewrite(text)
            left = sum(1 for ln in text.split("\n") if pk_erratum_is_vulnerable(ln))
            if left:
                raise RuntimeError(f"{src}: {left} vulnerable packed-fp32 instruction(s) survived the rewrite")
            open(asm, "w").write(text)
        _run([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm, "-o", dev_o], verbose, f"assembler on {src}")
        _run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", co, dev_o], verbose, f"lld on {src}")
        _run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
              "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", f"-input={co}", f"-output={fb}"], verbose, f"bundler on {src}")
        _run([hipcc] + flags + ["-x", "hip", "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", path, "-o", obj], verbose, f"hipcc (host) on {src}")
        return n
    finally:
        for f in (asm, dev_o, co, fb):
            if os.path.exists(f):
                os.remove(f)


def _build(verbose, more, rewrite):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("LAE_BUILD_EXTRA_FLAGS", "").split() + more      # probes only (e.g. -DLAE_GRID_STAMPS, tools/grid_bwd_stamps.py)
    if os.environ.get("LAE_BUILD_NO_PK_REWRITE") == "1":                    # probes only: the compiler's own code (tools/grid_loop_fault.sh)
        rewrite = False
    os.makedirs(LIBDIR, exist_ok=True)
    tag = f"{os.getpid()}.{abs(hash(SO)) % 100000}"                         # per-process, per-target object names: concurrent builds do not share files
    objs = [os.path.join(LIBDIR, f"{src.rsplit('.', 1)[0]}.{tag}.o") for src in SOURCES]
    rewritten = {}
    try:
        with concurrent.futures.ThreadPoolExecutor(len(SOURCES)) as ex:
            futs = {src: ex.submit(_compile_one, src, obj, FLAGS + extra, verbose, rewrite) for src, obj in zip(SOURCES, objs)}
            for src, f in futs.items():
                rewritten[src] = f.result()
        tmp_so = SO + f".{os.getpid()}.tmp"                      # link to a temp name, then rename: a reader never maps a half-written library
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_so] + objs)
        os.replace(tmp_so, SO)
        with open(SO + ".isa.json", "w") as f:
            json.dump({"pk_erratum_rewrite": bool(rewrite), "instructions_rewritten": rewritten, "total": sum(rewritten.values())}, f)
    finally:
        for o in objs:
            if os.path.exists(o):
                os.remove(o)
    return SO


if __name__ == "__main__":
    if "--out" in sys.argv:                                  # python -m laenerf_amd.build --out path.so [-DFLAG ...] [--no-pk-rewrite]
        print(build(out=sys.argv[sys.argv.index("--out") + 1], extra_flags=[a for a in sys.argv[1:] if a.startswith("-D")], verbose=True,
                    rewrite="--no-pk-rewrite" not in sys.argv))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
