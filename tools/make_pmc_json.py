#!/usr/bin/env python
"""profiles/pmc_traffic.json from the round's rocprofv3 runs (read by bench.py for roofline.traffic):

    python tools/make_pmc_json.py <pmc FETCH_SIZE dir> <pmc WRITE_SIZE dir> <kernel-trace dir> <batch> <tag> > profiles/pmc_traffic.json

HBM bytes per step and per launch of the dominant kernel class (every plain dW GEMM: gemm_tn_kernel<bf16_t, *, 0>,
gemm_tn_skinny_kernel<...>, gemm_tn128_kernel<...>) from the two PMC passes — units and the gfx950 FETCH_SIZE doubling as in tools/pmc_summary.py — and the
class's average launch duration from the kernel trace of the bench command."""
import collections
import csv
import glob
import json
import sys


def counter(d, name):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(glob.glob(f"{d}/*counter_collection.csv")[0])):
        if r["Counter_Name"] == name:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"]) * 1024.0
            cnt[r["Kernel_Name"]] += 1
    return tot, cnt


def dominant(n):
    return "gemm_tn_kernel<bf16_t, 0, 0>" in n or "gemm_tn_kernel<bf16_t, 1, 0>" in n or "gemm_tn_skinny_kernel" in n or "gemm_tn128_kernel" in n


fetch, cnt = counter(sys.argv[1], "FETCH_SIZE")
write, _ = counter(sys.argv[2], "WRITE_SIZE")
steps = max(1, sum(v for k, v in cnt.items() if "nchw_to_nhwc" in k))
rd, wr = 2.0 * sum(fetch.values()) / steps, sum(write.values()) / steps
dk_n = sum(v for k, v in cnt.items() if dominant(k))
dk_b = sum(2.0 * fetch[k] + write.get(k, 0.0) for k in cnt if dominant(k))
dur, n = 0.0, 0
for r in csv.DictReader(open(glob.glob(f"{sys.argv[3]}/*kernel_trace.csv")[0])):
    if dominant(r["Kernel_Name"]):
        dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        n += 1
print(json.dumps({
    "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, two separate passes of `bench.py --steps 2 --warmup 1 --no-kernel-probe "
              f"--no-cpu-baseline` (batch {sys.argv[4]}, 1x MI355X, {sys.argv[5]} build); FETCH_SIZE doubled per the gfx950 correction of "
              f"MI355X_MICROARCH.md, WRITE_SIZE as exported; tools/make_pmc_json.py",
    "step": {"read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr, "images": int(sys.argv[4]), "steps_profiled": steps},
    "dominant_kernel": {"name": "gemm_tn_kernel<bf16_t, *, 0> + gemm_tn_skinny_kernel<...> + gemm_tn128_kernel<...> (every plain dW GEMM of the step)",
                        "launches_per_step": dk_n / steps, "bytes_per_launch": dk_b / max(dk_n, 1),
                        "rocprof_avg_ms": dur / max(n, 1), "rocprof_launches": n},
}, indent=1))
