"""Build libsatrn_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsatrn_hip.so")
SOURCES = ["kernels_gemm.hip", "kernels_gemm_big.hip", "kernels_gemm_tall.hip", "kernels_elem.hip", "kernels_attn.hip", "kernels_attn2.hip", "kernels_encattn.hip", "kernels_se.hip", "kernels_mbconv.hip", "kernels_ar.hip", "kernels_decode.hip", "kernels_image.hip", "kernels_swin.hip", "engine.cpp", "satrn_abi.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "satrn_hip.h"))
    hstamp = max(os.path.getmtime(h) for h in headers)
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if force or newer(sp, obj) or hstamp > os.path.getmtime(obj):
            cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if procs or not os.path.exists(OUT):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
