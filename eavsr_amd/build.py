"""Build libeavsr_hip.so in-tree with hipcc for gfx950 (no GPU needed: hipcc cross-compiles).

    python -m eavsr_amd.build [--force] [--verbose] [--lab]

--lab (or EAVSR_BUILD_LAB=1) builds eavsr_amd/lib/libeavsr_lab.so: the product library's sources plus the retired schedules kept
for A/B measurements (LAB_SOURCES below); the product library is not touched.  EAVSR_LIB_PATH selects the file that is loaded.

The shared object lands in eavsr_amd/lib/ (git-ignored, but it travels with gpurun snapshots).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = LIB_PRODUCT = os.path.join(LIBDIR, "libeavsr_hip.so")
LIB_LAB = os.path.join(LIBDIR, "libeavsr_lab.so")
OBJDIR = OBJDIR_PRODUCT = os.path.join(LIBDIR, "obj")
ARCH = "gfx950"
# -fno-slp-vectorize: the SLP vectorizer packs pairs of independent fp32 additions / multiplications into v_pk_add_f32 /
# v_pk_mul_f32, which beside MFMAs are slower than the two plain instructions (MI355X_MICROARCH.md, "price of one filler beside
# MFMAs").  Measured: DCNv2 84-88 -> 77-78 us per 2 x 64 x 180 x 320 launch; the whole library 219.3 -> 217.7 ms per step
# (tools/visits/r4_d.sh, r4_t.sh).  The packed instructions the kernels WANT are written as inline assembly (conv_wino6.hip).
FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-ffp-contract=fast", "-fno-slp-vectorize"]


# per-source additions (none at present)
PER_FILE_FLAGS = {}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the native library cannot be built")


_HIPCC_VERSION = None


def hipcc_version() -> str:
    """`hipcc --version`, condensed (HIP version + the clang revision line): part of the build digest, because three kernels
    (conv_wino6.hip, conv_h16.hip, rcab_h16.hip) read their MFMA operands by `asm volatile` LDS reads with hand-counted
    `s_waitcnt` -- correct only as long as the compiler keeps the registers between a read and its wait where this toolchain
    keeps them.  A toolchain bump therefore rebuilds everything AND invalidates the guard library below, whose bit-compare
    (tests/test_hip_ops.py::test_winograd4_hand_counted_reads_equal_the_compilers) is the check that the assumption still holds."""
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        r = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True)
        lines = [ln.strip() for ln in (r.stdout or "").splitlines() if ln.startswith(("HIP version", "AMD clang version"))]
        _HIPCC_VERSION = " | ".join(lines) or "unknown hipcc"
    return _HIPCC_VERSION


# Product and lab.  The DEFAULT build holds what the default path and the documented modes use.  Schedules that
# were built, measured and retired -- kept because their A/B figures are quoted in DESIGN.md / docs/history -- compile only with
# `python -m eavsr_amd.build --lab` (-DEAVSR_LAB=1): whole files below, and the `#if EAVSR_LAB` pieces of predictor.hip (one kernel
# per pyramid level), conv_wino6.hip (channel-attention prologue inside the F(4x4,3x3) kernel) and dcnv2_x9.hip (the round-1 NCHW
# DCNv2 with nine partial products; its weight packing stays: eavsr_dcnv2_il_f32 reads the same slabs).  The header groups the entry
# points the same way; ops.py raises LabBuildRequired for a lab-only mode on a default build; lab tests skip.
LAB_SOURCES = {
    "dcnv2_ws.hip",      # wave-specialised DCNv2 schedule (round 2): not faster at the alignment's offsets
    "conv_x9.hip",       # direct 3x3 convolution with all nine bf16 partial products
    "conv_wino.hip",     # Winograd F(2x2,3x3): 15 % slower end to end than F(4x4,3x3) (round 1)
    "rcab_h16.hip",      # conv -> ReLU -> conv of a 16-bit RCAB as one launch: bit-identical, slower in the step (round 5)
}


def sources(lab: bool = False):
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip") and (lab or f not in LAB_SOURCES))


def _digest(lab: bool = False) -> str:
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(PER_FILE_FLAGS.items())) + hipcc_version() + ("|lab" if lab else "")).encode())
    files = sources(lab) + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    files.append(os.path.join(os.path.dirname(HERE), "include", "eavsr_hip.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    return h.hexdigest()


def lab_requested() -> bool:
    """EAVSR_BUILD_LAB=1 makes every build of this process (the one `import eavsr_amd` triggers included) a lab build"""
    return os.environ.get("EAVSR_BUILD_LAB", "0") == "1"


def build_native(force: bool = False, verbose: bool = False, extra_flags=(), lab=None) -> str:
    lab = lab_requested() if lab is None else bool(lab)
    # the two flavours are two files: the lab build never replaces the product library (eavsr_amd/lib/libeavsr_lab.so, selected with
    # EAVSR_LIB_PATH -- or by EAVSR_BUILD_LAB=1, which _native.py honours as well)
    LIB = LIB_LAB if lab else LIB_PRODUCT
    OBJDIR = OBJDIR_PRODUCT + ("_lab" if lab else "")
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build_lab.stamp" if lab else "build.stamp")
    dig = _digest(lab) + "|" + " ".join(extra_flags)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = _hipcc()
    extra_flags = (*extra_flags, f"-DEAVSR_LAB={1 if lab else 0}")
    for f in os.listdir(OBJDIR):      # objects of sources that are no longer part of the flavour must not be linked
        os.remove(os.path.join(OBJDIR, f))

    def compile_one(src):
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc, *FLAGS, *PER_FILE_FLAGS.get(os.path.basename(src), []), *extra_flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources(lab)))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


GUARD_DIR = os.path.join(LIBDIR, "guard")
# name -> (sources, extra flags): the product kernel with its hand-scheduled pieces handed back to the compiler
GUARD_VARIANTS = {
    "wino4_creads": (["conv_wino6.hip", "capi.hip"], ["-DEAVSR_W4_COMPILER_READS"]),
}


def build_guard(name: str = "wino4_creads", force: bool = False) -> str:
    """eavsr_amd/lib/guard/lib<name>.so: conv_wino6.hip built with -DEAVSR_W4_COMPILER_READS (every LDS operand read and wait
    left to the compiler) beside capi.hip -- the reference the toolchain guard compares the product kernel with, bit for bit, on
    the GPU.  Same flags as the product build; keyed on the same digest (sources, flags, hipcc version)."""
    srcs, extra = GUARD_VARIANTS[name]
    os.makedirs(GUARD_DIR, exist_ok=True)
    out = os.path.join(GUARD_DIR, f"lib{name}.so")
    stamp = out + ".stamp"
    dig = _digest() + "|" + " ".join(extra)
    extra = [*extra, "-DEAVSR_LAB=0"]
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == dig:
        return out
    cmd = [_hipcc(), *FLAGS, *extra, "-shared", *[os.path.join(CSRC, f) for f in srcs], "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on the guard library {name}:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    return out


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose="--verbose" in sys.argv, lab=("--lab" in sys.argv) or lab_requested()))
