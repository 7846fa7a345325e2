"""Build libeavsr_hip.so in-tree with hipcc for gfx950 (no GPU needed: hipcc cross-compiles).

    python -m eavsr_amd.build [--force] [--verbose]

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
LIB = os.path.join(LIBDIR, "libeavsr_hip.so")
OBJDIR = os.path.join(LIBDIR, "obj")
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


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest() -> str:
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(PER_FILE_FLAGS.items()))).encode())
    files = sources() + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    files.append(os.path.join(os.path.dirname(HERE), "include", "eavsr_hip.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    return h.hexdigest()


def build_native(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.stamp")
    dig = _digest() + "|" + " ".join(extra_flags)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = _hipcc()

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
        objs = list(ex.map(compile_one, sources()))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
