"""Builds the gfx950 shared library (C ABI in include/mia_letkf.h) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU; the resulting .so is git-ignored but
travels with the working tree to the GPU box.
"""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmia_letkf.so")
STAMP = os.path.join(LIB_DIR, "libmia_letkf.stamp")
SOURCES = ["localize.hip", "letkf_entry.hip", "etkf_global.hip", "letkf_wave.hip", "letkf_sys.hip", "letkf_cheb.hip", "letkf_tile.hip", "letkf_tile_split.hip", "letkf_tile2.hip", "letkf_tile2w.hip", "letkf_tile2p.hip", "letkf_tile2f.hip", "lketkf_tile.hip", "tile_lists.hip", "sharded_step.hip", "obs_space.hip", "ienks.hip", "apply_local.hip", "api.cc"]
HEADERS = ["mia_common.h", "mia_options.h", "mia_jacobi.h", "mia_jacobi_sym.h", "mia_kernel_prog.h", "mia_localize_dev.h", "mia_kernels.h", "mia_pack_dev.h", "mia_tiles.h", os.path.join(ROOT, "include", "mia_letkf.h")]
INCLUDES = {"letkf_tile_split.hip": ["letkf_tile.hip"]}     # sources that include another source
# per-source flags.  letkf_tile2.hip: no SLP vectoriser -- it packs the recurrence's scalar f32 multiply-adds into v_pk_fma_f32,
# which cost more than two v_fma_f32 beside MFMAs (MI355X_MICROARCH.md, 'price of one filler beside MFMAs')
SOURCE_FLAGS = {"letkf_tile2.hip": ["-fno-slp-vectorize"], "letkf_tile2w.hip": ["-fno-slp-vectorize"], "letkf_tile2p.hip": ["-fno-slp-vectorize"],
                "letkf_tile2f.hip": ["-fno-slp-vectorize"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _flags():
    """Compile flags; MIA_BUILD_FLAGS (read when build() runs) adds defines for the A/B and experiment builds of tools/."""
    return FLAGS + os.environ.get("MIA_BUILD_FLAGS", "").split()


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _digest():
    h = hashlib.sha256()
    for f in _sources() + [f if os.path.isabs(f) else os.path.join(CSRC, f) for f in HEADERS]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(_flags()).encode())
    return h.hexdigest()


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _object_digest(src):
    h = hashlib.sha256()
    for f in [src] + [f if os.path.isabs(f) else os.path.join(CSRC, f) for f in HEADERS + INCLUDES.get(os.path.basename(src), [])]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(_flags() + SOURCE_FLAGS.get(os.path.basename(src), [])).encode())
    return h.hexdigest()[:24]


def _compile(job):
    src, obj, verbose = job
    cmd = ([hipcc_path()] + [f for f in _flags() if f != "-shared"] + SOURCE_FLAGS.get(os.path.basename(src), []) +
           ["-c", src, "-o", obj + ".tmp"])
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s%s" % (os.path.basename(src), res.stdout, res.stderr))
    os.replace(obj + ".tmp", obj)
    return obj


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into lib/libmia_letkf.so (skipped if up to date).  One object per
    source, cached under lib/obj by content digest and compiled in parallel, then one link."""
    os.makedirs(LIB_DIR, exist_ok=True)
    dig = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == dig:
                return LIB_PATH
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(obj_dir, "%s.%s.o" % (os.path.basename(src), _object_digest(src)))
        objs.append(obj)
        if force or not os.path.exists(obj):
            jobs.append((src, obj, verbose))
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(_compile, jobs))
    for f in os.listdir(obj_dir):                      # drop objects of older source versions
        if os.path.join(obj_dir, f) not in objs:
            os.remove(os.path.join(obj_dir, f))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-ldl", "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
