"""Build a VARIANT of libcvnets_hip.so for same-box A/B runs (CVNETS_HIP_LIB): one source recompiled with extra -D flags, linked with the
objects of the regular build.      python tools/build_variant.py NAME SOURCE.hip[,SOURCE2.hip...] -DFOO=1 [-DBAR=2 ...]  ->  ml-cvnets_amd/lib/libcvnets_hip_NAME.so"""
import importlib.util
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("b", os.path.join(REPO, "ml-cvnets_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)


def main():
    name, srcs, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    b.build()
    new = []
    for src in srcs:
        obj = os.path.join(b.BUILD, src.replace(".hip", f"_{name}.o"))
        cmd = [b.hipcc()] + b.FLAGS + (["-munsafe-fp-atomics"] if src in b.UNSAFE_FP_ATOMICS else []) + (["-DCVH_NT_LOADS"] if src in b.NT_LOADS else []) + flags + ["-c", os.path.join(b.CSRC, src), "-o", obj]
        subprocess.check_call(cmd)
        new.append(obj)
    objs = [os.path.join(b.BUILD, s.replace(".hip", ".o")) for s in b.SOURCES if s not in srcs] + new
    out = os.path.join(b.LIBDIR, f"libcvnets_hip_{name}.so")
    subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
    print(out)


if __name__ == "__main__":
    main()
