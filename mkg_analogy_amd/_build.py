"""Build libmart_hip.so (gfx950 only) in-tree with hipcc.  No JIT cache: the .so travels with the repo snapshot."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmart_hip.so")
SOURCES = ["util.hip", "gemm_nt.hip", "gemm_tn.hip", "norm_embed.hip", "attention.hip", "fusion.hip", "head_optim.hip", "precise.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# per-file extras: attention keeps MFMA results in arch VGPRs (the softmax consumes them with VALU right away; the
# default AGPR form cost ~150 v_accvgpr moves per 16 MFMAs)
EXTRA = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "precise.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wno-unused-result", "-Wno-pass-failed"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(ROOT, "include", "mart_hip.h")]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return obj

    if jobs:
        if verbose:
            print(f"[mart build] compiling {len(jobs)} file(s) for gfx950", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        if verbose:
            print(f"[mart build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
