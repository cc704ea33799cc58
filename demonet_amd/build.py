"""Builds demonet_amd/lib/libdemonet_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m demonet_amd.build [--force]

One object per .hip file (postprocess.hip with -ffp-contract=off: decode/IoU must round like the reference),
linked into one C-ABI shared library. In-tree output so the .so travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdemonet_hip.so")
SOURCES = ["plan.hip", "pointwise.hip", "depthwise.hip", "dense.hip", "postprocess.hip", "expdw.hip", "tail.hip", "convbig.hip", "pwdirect.hip", "loss.hip", "headfuse.hip"]
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs, not AGPRs -- the small-tile kernels otherwise spend a v_accvgpr_read per
# accumulator value on the way to their epilogues (not for convbig.hip: its 256 x 256 tiles need the AGPR half of the file)
VGPR_MFMA = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
EXTRA = {"postprocess.hip": ["-ffp-contract=off"], "pwdirect.hip": VGPR_MFMA, "pointwise.hip": VGPR_MFMA, "tail.hip": VGPR_MFMA, "depthwise.hip": VGPR_MFMA}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
          "-fvisibility=hidden", "-fgpu-rdc" if False else "-fno-gpu-rdc"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_stamps(verbose=True):
    """Dev build: lib/libdemonet_hip_stamps.so = the same objects with every kernel file that has probe hooks compiled with -DDN_DEV_STAMPS
    (per-workgroup phase stamps and the dn_debug_*_stamps / dn_debug_pw_tile exports: tools/probe_*.py, tools/tune_pw.py). Load it with
    DEMONET_HIP_LIB=<path>; the product library carries neither the stamps nor those exports."""
    build(verbose=verbose)
    stamped = ["headfuse.hip", "expdw.hip", "pointwise.hip", "depthwise.hip", "postprocess.hip", "tail.hip", "convbig.hip"]
    objs = [os.path.join(LIBDIR, s.replace(".hip", ".o")) for s in SOURCES if s not in stamped]
    for src in stamped:
        obj = os.path.join(LIBDIR, src.replace(".hip", "_stamps.o"))
        cmd = [HIPCC] + COMMON + EXTRA.get(src, []) + ["-DDN_DEV_STAMPS", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    out = os.path.join(LIBDIR, "libdemonet_hip_stamps.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(os.path.dirname(HERE), "include", "demonet_hip.h"),
               os.path.join(os.path.dirname(HERE), "include", "demonet_hip_debug.h")]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            cmd = [HIPCC] + COMMON + EXTRA.get(src, []) + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # a library that links but does not load (an unresolved kernel stub) must fail HERE, not on the GPU box
    import ctypes
    ctypes.CDLL(LIB)
    return LIB


if __name__ == "__main__":
    print(build_stamps() if "--stamps" in sys.argv else build(force="--force" in sys.argv))
