"""Build the gfx950 shared library in-tree (``livelyspeaker_amd/libls_hip.so``).

``hipcc`` cross-compiles for gfx950 without a GPU, so this runs in the build container; the built
``.so`` is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libls_hip.so")
SOURCES = ["ls_api.cpp", "ls_torch_rng.cpp", "ls_sag_api.cpp", "ls_step.hip", "ls_step_beat.hip", "ls_coop.hip", "ls_pass.hip", "ls_mix.hip", "ls_long.hip", "ls_prepare.hip", "ls_sag.hip", "ls_post.hip", "ls_conv.hip", "ls_train_api.cpp", "ls_gemm.hip", "ls_train_kernels.hip", "ls_train_bwd.hip", "ls_eval.hip"]
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "ls_hip.h")]
STAMP = LIB + ".srchash"          # hash of the sources the library was built from (travels with it; git-ignored)


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0)")


def source_hash() -> str:
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def is_stale() -> bool:
    """Content-based when the build left its stamp (file times do not survive every copy of the tree); mtimes otherwise."""
    if not os.path.exists(LIB):
        return True
    if os.path.exists(STAMP):
        with open(STAMP) as f:
            return f.read().strip() != source_hash()
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def _object_for(hipcc, src, flags, objdir):
    """Compile one source to a cached object: the key covers the source, every header and the flags."""
    h = hashlib.sha256(" ".join(flags).encode())
    for path in [os.path.join(CSRC, src)] + HEADERS:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    obj = os.path.join(objdir, f"{os.path.splitext(src)[0]}.{h.hexdigest()[:16]}.o")
    if os.path.exists(obj):
        return obj, None
    tmp = f"{obj}.tmp{os.getpid()}"
    res = subprocess.run([hipcc, "-c"] + flags + [os.path.join(CSRC, src), "-o", tmp], capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        return None, f"{src}:\n{res.stdout}{res.stderr}"
    os.replace(tmp, obj)
    for stale in glob.glob(os.path.join(objdir, f"{os.path.splitext(src)[0]}.*.o")):     # one cached object per source
        if stale != obj:
            os.remove(stale)
    return obj, None


def build_library(force: bool = False, verbose: bool = False, defines=(), out: str = None, extra_flags=()) -> str:
    """Compile every source for gfx950 (one hipcc per file, in parallel, objects cached under build/obj) and link the
    shared library.  defines/out: build an A/B or debug variant (e.g. defines=['LS_DEBUG'], out='build/variants/debug.so');
    force=True ignores the object cache."""
    from concurrent.futures import ThreadPoolExecutor
    LIB = out or globals()['LIB']
    if not force and out is None and not is_stale():
        return LIB
    hipcc = hipcc_path()
    # -fno-slp-vectorize: SLP packs adjacent scalar f32 FMAs into v_pk_fma_f32 + v_mov shuffles, which is slower than
    # the scalar form next to MFMAs (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
             "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + ["-D" + d for d in defines] + list(extra_flags)
    variant = list(defines) + list(extra_flags)
    objdir = os.path.join(ROOT, "build", "obj" + ("-" + hashlib.sha256(" ".join(variant).encode()).hexdigest()[:8] if variant else ""))
    os.makedirs(objdir, exist_ok=True)
    if force:
        for o in glob.glob(os.path.join(objdir, "*.o")):
            os.remove(o)
    digest = source_hash()
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        results = list(ex.map(lambda s: _object_for(hipcc, s, flags, objdir), SOURCES))
    errors = [e for _, e in results if e]
    if errors:
        raise RuntimeError("hipcc failed:\n" + "\n".join(errors))
    tmp = f"{LIB}.tmp{os.getpid()}"              # several ranks may find the library stale at once: private temp, atomic rename
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _ in results] + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB)
    if out is None and not defines and not extra_flags:
        with open(STAMP + f".tmp{os.getpid()}", "w") as f:
            f.write(digest + "\n")
        os.replace(STAMP + f".tmp{os.getpid()}", STAMP)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
