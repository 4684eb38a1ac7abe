"""Build libesahrnet.so in-tree with hipcc for gfx950 (the .so travels to the GPU box with the
repo snapshot; nothing is JIT-compiled at run time)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libesahrnet.so")
SOURCES = ["conv_mfma.hip", "conv_s2c32.hip", "conv_x6.hip", "conv1x1.hip", "stem.hip", "stem_fused.hip", "bblock32.hip", "cbam.hip", "crops.hip", "fuse.hip", "head.hip", "head_fused.hip", "head_fused2.hip", "head_fused_bf.hip", "head_gather.hip", "head_t.hip", "head_x6.hip", "keypoints.hip", "layout.hip", "plan.hip", "pnp_host.hip"]
# conv_x6.hip: MFMA results that a VALU instruction reads next (the per-row fresh sums) are allocated in VGPRs — with
# AGPR destinations hipcc copies them out right behind the chain's last MFMA and pads the hazard with s_nop (csrc/conv_x6.hip)
PER_FILE_FLAGS = {"conv_x6.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
HEADERS = ["kernels.h", "sb.h", "conv_cfg.h", "devstate.h", os.path.join("..", "..", "include", "esahrnet.h")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libesahrnet.so)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
    flags += os.environ.get("ESA_HIPCC_FLAGS", "").split()       # tuning experiments only
    procs = []
    for s in SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [_hipcc(), *flags, *PER_FILE_FLAGS.get(s, []), "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
