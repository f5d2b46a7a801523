"""Build laenerf_amd/lib/liblaenerf_hip.so (HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU; the resulting .so is kept in-tree (git-ignored) so it
travels to the GPU box with the snapshot.  `python -m laenerf_amd.build [--force]`.
"""
import os
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


def _deps():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    out.append(os.path.join(HERE, "..", "include", "laenerf.h"))
    return out


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("LAE_BUILD_EXTRA_FLAGS", "").split()      # probes only (e.g. -DLAE_GRID_STAMPS, tools/grid_bwd_stamps.py)
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, f"{src.rsplit('.', 1)[0]}.{os.getpid()}.o")   # per-process object names: concurrent builds do not share files
        objs.append(obj)
        cmd = [hipcc] + FLAGS + extra + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
        if verbose and out:
            print(out.decode())
    tmp_so = SO + f".{os.getpid()}.tmp"                      # link to a temp name, then rename: a reader never maps a half-written library
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_so] + objs
    subprocess.check_call(cmd)
    os.replace(tmp_so, SO)
    for o in objs:
        os.remove(o)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
