"""Builds imagemosaicing_amd/libmi355mosaic.so (hand-written HIP for gfx950 + host C++) in-tree.

    python -m imagemosaicing_amd.build [--force]

hipcc cross-compiles gfx950 without a GPU.  -ffp-contract=off is part of the numerical contract: the
warp / RANSAC kernels must round every product and sum separately like the reference compiled by g++
(fused operations appear only where the source calls fmaf explicitly).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libmi355mosaic.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"] + os.environ.get("MI355_EXTRA_HIPCC_FLAGS", "").split()   # kernel A/B builds (scratch/)


# per-file extra flags.  ransac.hip: the SLP vectoriser packs the unrolled 8 x 8 solves into v_pk_mul/add_f32 and pays for it with ~480
# register moves and scratch traffic per Gauss-Newton step; a packed f32 instruction occupies the SIMD as long as its two halves issued
# one by one (profiles/r04_pk_rate.txt), so the moves are pure loss: 557 -> 454 us of polish per pair.  The one loop that gains from
# pairs (the support count) is written with explicit two-element vectors.
PER_FILE = {"ransac.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


GEN = os.path.join(CSRC, "gen_blur16_asm.py")          # writes blur16_asm.inc (the hand-scheduled row loop of blur16_stream)
INC = os.path.join(OBJ, "blur16_asm.inc")              # generated text lives with the objects (the source directory stays as checked out); found through -I


def generate():
    os.makedirs(OBJ, exist_ok=True)
    stale = os.path.join(CSRC, "blur16_asm.inc")           # where rounds 3-4 wrote it: never regenerated any more, must not shadow the fresh one
    if os.path.exists(stale):
        os.remove(stale)
    if not os.path.exists(INC) or os.path.getmtime(INC) < os.path.getmtime(GEN):
        out = subprocess.run([sys.executable, GEN], capture_output=True, text=True, check=True).stdout
        with open(INC, "w") as f:
            f.write(out)


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if os.path.exists(INC):
        hs.append(INC)
    hs.append(os.path.join(os.path.dirname(HERE), "include", "mi355_mosaic.h"))
    return hs


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(INC):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in sources() + headers() + [GEN, os.path.abspath(__file__)])


def compile_one(src):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    newest = max(os.path.getmtime(p) for p in [src] + headers())
    if os.path.exists(obj) and os.path.getmtime(obj) > newest:
        return obj
    cmd = [HIPCC] + FLAGS + ["-I", OBJ] + PER_FILE.get(os.path.basename(src), []) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False):
    if not force and not needs_build():
        return LIB
    if not os.path.exists(HIPCC):
        if os.path.exists(LIB):
            return LIB                      # GPU box without a toolchain change: use the prebuilt library
        raise RuntimeError("hipcc not found and no prebuilt libmi355mosaic.so")
    os.makedirs(OBJ, exist_ok=True)
    generate()
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
