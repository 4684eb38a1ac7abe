"""Build libesahrnet.so in-tree with hipcc for gfx950 (the .so travels to the GPU box with the
repo snapshot; nothing is JIT-compiled at run time)."""
from __future__ import annotations

import json
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libesahrnet.so")
USAGE = os.path.join(HERE, "build", "resource_usage.json")      # per kernel: VGPRs, scratch bytes, spills, waves per SIMD (hipcc's remarks)
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
    flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
             "-Rpass-analysis=kernel-resource-usage"]
    flags += os.environ.get("ESA_HIPCC_FLAGS", "").split()       # tuning experiments only
    procs = []
    for s in SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [_hipcc(), *flags, *PER_FILE_FLAGS.get(s, []), "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    usage = {}
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        usage.update(_parse_usage(s, out))
        rest = "\n".join(ln for ln in out.splitlines() if "warning:" in ln or "error:" in ln)
        if verbose and rest.strip():
            print(rest, file=sys.stderr)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(LIB + ".tmp", LIB)
    with open(USAGE, "w") as f:
        json.dump(usage, f, indent=1, sort_keys=True)
    return LIB


def _parse_usage(src: str, out: str) -> dict:
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {"<file>:<mangled kernel>": {...}} (tests/test_host_cpu.py reads
    it: the hot kernels must not touch scratch memory — a struct copied through it or an array hipcc could not promote cost
    the fused head 35 % — and must keep the occupancy their design counts on)."""
    res, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "VGPRs Spill": "vgpr_spill",
            "SGPRs Spill": "sgpr_spill", "Occupancy [waves/SIMD]": "waves_per_simd", "LDS Size [bytes/block]": "lds"}
    for ln in out.splitlines():
        m = re.search(r"remark: .*?Function Name: (\S+)", ln)
        if m:
            cur = res.setdefault(f"{src}:{m.group(1)}", {})
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z /\[\]]+?): (\d+)", ln)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return res


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
