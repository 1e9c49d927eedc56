"""Compressed trace of memory operations / waits / branches of one kernel's gfx950 ISA (developer tool): where do the loads sit relative
to the waits that cover them?   usage: python tools/isa_trace.py <file.hip> <mangled-name substring> [max chars]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "ml-cvnets_amd", "csrc")
src = os.path.join(CSRC, sys.argv[1])
out = f"/tmp/{sys.argv[1]}.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{REPO}/include", f"-I{CSRC}", "-S", "--cuda-device-only", src, "-o", out],
               check=True, stderr=subprocess.DEVNULL)
for f in re.split(r"\n(?=_Z[\w]+:)", open(out).read()):
    name = f.split(":")[0]
    if not name.startswith("_Z") or sys.argv[2] not in name:
        continue
    lines = [l.strip() for l in f.splitlines() if l.strip() and not l.strip().startswith((";", "."))]
    seq, alu = [], 0
    for l in lines:
        op = l.split()[0]
        if op.startswith(("global_load", "global_store", "global_atomic", "s_waitcnt", "s_barrier", "ds_", "v_mfma", "s_cbranch", "buffer_", "flat_", "scratch_")):
            if alu:
                seq.append(f"[{alu}]")
                alu = 0
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", l)
                g = re.search(r"lgkmcnt\((\d+)\)", l)
                op = "WAIT" + (f" vm{m.group(1)}" if m else "") + (f" lgkm{g.group(1)}" if g else "")
            seq.append(op)
        else:
            alu += 1
    comp, prev, n = [], None, 0
    for o in seq + [None]:
        if o == prev:
            n += 1
        else:
            if prev is not None:
                comp.append(prev + (f" x{n}" if n > 1 else ""))
            prev, n = o, 1
    print(name, len(lines), "instructions")
    print(" | ".join(comp)[: int(sys.argv[3]) if len(sys.argv) > 3 else 6000])
    print()
