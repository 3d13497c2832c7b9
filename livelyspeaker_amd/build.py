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
SOURCES = ["ls_api.cpp", "ls_sag_api.cpp", "ls_step.hip", "ls_step_seq.hip", "ls_prepare.hip", "ls_gemm.hip", "ls_sag.hip", "ls_post.hip", "ls_conv.hip", "ls_train_api.cpp", "ls_train_gemm.hip", "ls_train_kernels.hip", "ls_train_bwd.hip", "ls_eval.hip"]
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


def build_library(force: bool = False, verbose: bool = False, defines=(), out: str = None) -> str:
    """defines/out: build an A/B variant (e.g. defines=['LS_GRP=2'], out='build/variants/grp2.so')."""
    LIB = out or globals()['LIB']
    if not force and out is None and not is_stale():
        return LIB
    # -fno-slp-vectorize: SLP packs adjacent scalar f32 FMAs into v_pk_fma_f32 + v_mov shuffles, which is slower than
    # the scalar form next to MFMAs (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + ["-D" + d for d in defines]
    tmp = f"{LIB}.tmp{os.getpid()}"              # several ranks may find the library stale at once: private temp, atomic rename
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    digest = source_hash()
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB)
    if out is None and not defines:
        with open(STAMP + f".tmp{os.getpid()}", "w") as f:
            f.write(digest + "\n")
        os.replace(STAMP + f".tmp{os.getpid()}", STAMP)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
