"""Per-kernel histogram of `s_waitcnt vmcnt(N)` in the gfx950 ISA of the HIP sources: a kernel that software-prefetches but only ever
waits with vmcnt(0) has lost its prefetch distance (the compiler could not count the outstanding loads, typically because they sit
behind per-lane branches) and runs at one memory latency per stage.   usage: python tools/waitcnt_survey.py [file.hip ...]"""
import collections
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "ml-cvnets_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
for fn in files:
    src = os.path.join(CSRC, os.path.basename(fn))
    out = f"/tmp/{os.path.basename(fn)}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{REPO}/include", f"-I{CSRC}", "-S", "--cuda-device-only",
                    src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    for f in re.split(r"\n(?=_Z[\w]+:)", open(out).read()):
        name = f.split(":")[0]
        if not name.startswith("_Z"):
            continue
        loads = len(re.findall(r"(?:global|buffer)_load_dwordx[24]", f))
        if loads < 6:
            continue
        w = collections.Counter(re.findall(r"s_waitcnt vmcnt\((\d+)\)", f))
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem)[:72]
        print(f"{dem:74s} loads {loads:3d}  vmcnt {sorted(w.items(), key=lambda kv: int(kv[0]))}")
