"""Build the gfx950 shared library in-tree (``livelyspeaker_amd/libls_hip.so``).

``hipcc`` cross-compiles for gfx950 without a GPU, so this runs in the build container; the built
``.so`` is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libls_hip.so")
SOURCES = ["ls_api.cpp", "ls_sag_api.cpp", "ls_step.hip", "ls_step_seq.hip", "ls_prepare.hip", "ls_gemm.hip", "ls_sag.hip", "ls_post.hip", "ls_conv.hip", "ls_train_api.cpp", "ls_train_gemm.hip", "ls_train_kernels.hip", "ls_train_bwd.hip", "ls_eval.hip"]
HEADERS = [os.path.join(CSRC, "ls_internal.h"), os.path.join(CSRC, "ls_philox.h"), os.path.join(CSRC, "ls_step_common.h"),
           os.path.join(ROOT, "include", "ls_hip.h")]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
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
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
