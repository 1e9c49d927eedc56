#!/usr/bin/env python
"""Builds libcvnets_hip.so (gfx950) from csrc/*.hip with plain hipcc — no torch extension machinery:
the library is a C-ABI .so (include/cvnets_hip.h) loaded through ctypes.

    python ml-cvnets_amd/build.py [--force] [--verbose]

Objects are cached by source mtime under ml-cvnets_amd/build/; the .so lands in ml-cvnets_amd/lib/
(git-ignored, but it travels to the GPU box with the gpurun snapshot)."""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
BUILD = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcvnets_hip.so")
SOURCES = ["gemm.hip", "gemm_fx.hip", "gemm_big.hip", "gemm_stream.hip", "gemm_rows.hip", "conv3x3.hip", "conv3x3_dw.hip", "stem.hip", "ir_bwd.hip", "ir_fwd.hip", "elementwise.hip", "dwconv.hip", "dwfused.hip", "dwx.hip", "dwxs.hip", "bngram.hip", "ir_pb.hip", "bnlink.hip", "layernorm.hip", "attention.hip", "tokens.hip", "linattn.hip", "optim.hip", "comm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + CSRC, "-I" + INC]
# hardware float atomics (instead of compare-and-swap loops) only where a kernel issues a float atomicAdd at all — none of them runs in the
# MobileViT / ViT training step (reproducibility): the token-embedding gradient of CLIP (tokens.hip), the token-axis
# LayerNorm quirk branch (layernorm.hip), the scratch-less dW mode nothing calls (gemm_tn.hpp, included by gemm.hip)
UNSAFE_FP_ATOMICS = {"tokens.hip", "layernorm.hip", "gemm.hip"}
# non-temporal 16-byte global LOADS (common.hpp: CVH_NT_LOADS) for the two files whose kernels are pure streams over tensors far larger than
# the caches (BatchNorm apply / backward / column reductions, dropout, LayerNorm): -0.5 ms per step same box (69.8 vs 70.3, twice).  The same
# hint on the operand loads of gemm_stream_kernel costs +1.7 ms (its 16-byte row pieces share cache lines: they need the L1) and +0.3 ms on the
# ir_* streaming kernels — measured in round 5 (gpurun_out/r05g), left off there (CVH_NT_STREAM).
NT_LOADS = {"elementwise.hip", "layernorm.hip", "bngram.hip"}


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def compile_one(src, force, verbose):
    obj = os.path.join(BUILD, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src), os.path.join(INC, "cvnets_hip.h"), __file__] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".hpp")]
    if not force and not _stale(obj, deps):
        return obj, 0.0, ""
    t0 = time.time()
    cmd = [hipcc()] + FLAGS + (["-munsafe-fp-atomics"] if src in UNSAFE_FP_ATOMICS else []) + (["-DCVH_NT_LOADS"] if src in NT_LOADS else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, time.time() - t0, r.stderr


def build(force=False, verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        res = list(ex.map(lambda s: compile_one(s, force, verbose), SOURCES))
    objs = [r[0] for r in res]
    for (o, dt, log), s in zip(res, SOURCES):
        if dt:
            print(f"[build] {s}: {dt:.1f}s")
        if verbose and log:
            print(log)
    if force or _stale(LIB, objs):
        r = subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        print(f"[build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    try:
        build(a.force, a.verbose)
    except RuntimeError as e:
        print(str(e)[-6000:])
        sys.exit(1)
