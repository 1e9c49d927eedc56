#!/usr/bin/env python
"""profiles/step_profile.json — the per-kernel-family picture of ONE bench.py training step, from the round's rocprofv3 runs of the bench
command itself (read by bench.py for `roofline.traffic` and `roofline.dominant_kernel`):

    python tools/make_step_profile.py <kernel-trace dir> <pmc FETCH_SIZE dir> <pmc WRITE_SIZE dir> <batch> <tag> > profiles/step_profile.json

* kernel trace (`rocprofv3 --kernel-trace --stats`): the last full step (between two launches of the step's first kernel: stem_fwd_kernel, or nchw_to_nhwc on builds without it): launches, total and average
  duration per kernel family (symbol name up to its template arguments), share of the summed kernel time, the DOMINANT family = largest share;
* PMC passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate runs): HBM bytes per step and per family — FETCH_SIZE doubled (gfx950: 128-B
  requests tallied at 64 B, MI355X_MICROARCH.md), WRITE_SIZE as exported;
* `src_hash`: SHA-1 over the kernel sources + host package + bench.py the numbers were measured on; bench.py refuses to quote a profile
  whose hash differs from the tree it runs in (it then reports its live measurements only and says so)."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src_hash():
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(REPO, "ml-cvnets_amd", "csrc", "*")) + glob.glob(os.path.join(REPO, "ml-cvnets_amd", "cvnets_amd", "*.py")) +
                   [os.path.join(REPO, "bench.py"), os.path.join(REPO, "include", "cvnets_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def family(name):
    n = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    return re.sub(r"<.*", "", n)


def counter(d, cname):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0])):
        if r["Counter_Name"] == cname:
            tot[family(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1024.0
            cnt[family(r["Kernel_Name"])] += 1
    return tot, cnt


def main():
    trace, d_f, d_w, batch, tag = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    rows = sorted(csv.DictReader(open(glob.glob(f"{trace}/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if ("stem_fwd_kernel" in r["Kernel_Name"] or "nchw_to_nhwc" in r["Kernel_Name"])]
    step = rows[idx[-2]:idx[-1]]
    window_ms = (int(rows[idx[-1]]["Start_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
    fam = collections.OrderedDict()
    for r in step:
        f = fam.setdefault(family(r["Kernel_Name"]), {"launches": 0, "ms": 0.0})
        f["launches"] += 1
        f["ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    total_ms = sum(f["ms"] for f in fam.values())
    fetch, cnt = counter(d_f, "FETCH_SIZE")
    write, _ = counter(d_w, "WRITE_SIZE")
    steps = max(1, sum(v for k, v in cnt.items() if "stem_fwd_kernel" in k or "nchw_to_nhwc" in k) or 1)
    out_f = []
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        b = (2.0 * fetch.get(name, 0.0) + write.get(name, 0.0)) / steps
        out_f.append({"family": name, "launches_per_step": f["launches"], "ms_per_step": round(f["ms"], 4), "avg_ms": round(f["ms"] / f["launches"], 5),
                      "share_of_kernel_time": round(f["ms"] / total_ms, 4), "hbm_bytes_per_step": b,
                      "hbm_GBps": round(b / (f["ms"] * 1e-3) / 1e9, 1) if f["ms"] > 0 else None})
    rd, wr = 2.0 * sum(fetch.values()) / steps, sum(write.values()) / steps
    print(json.dumps({
        "source": f"rocprofv3 --kernel-trace --stats and --pmc FETCH_SIZE / WRITE_SIZE (three separate runs) of `bench.py --no-kernel-probe --no-cpu-baseline` "
                  f"(batch {batch}, 1x MI355X, build {tag}); FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE as exported; "
                  f"tools/make_step_profile.py",
        "tag": tag, "src_hash": src_hash(),
        "step": {"images": batch, "window_ms": round(window_ms, 3), "kernel_time_ms": round(total_ms, 3), "launches": len(step), "read_bytes": rd, "write_bytes": wr,
                 "total_bytes": rd + wr, "pmc_steps_profiled": steps},
        "dominant": out_f[0]["family"],
        "families": out_f[:40],
    }, indent=1))


if __name__ == "__main__":
    main()
