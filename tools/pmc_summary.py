#!/usr/bin/env python
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately, as MI355X_MICROARCH.md prescribes) of
`bench.py` into per-kernel HBM traffic per launch and per training step.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/pmc_FETCH_SIZE -o pmc -- python bench.py --steps 2 --warmup 1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/pmc_WRITE_SIZE -o pmc -- python bench.py --steps 2 --warmup 1
    python tools/pmc_summary.py out/pmc_FETCH_SIZE out/pmc_WRITE_SIZE

Units / corrections (guide, HBM section): both counters are in KiB-sized units of 64-B fabric requests as exported by rocprofv3
(value x 1024 = bytes); on gfx950 FETCH_SIZE tallies the 128-B requests of wide streaming reads at 64 B, so it is DOUBLED here;
WRITE_SIZE is taken as is (uncalibrated — treat as a lower bound).  Steps = number of stem_fwd_kernel (older builds: nchw_to_nhwc) dispatches (one per step)."""
import collections
import csv
import glob
import re
import sys


def load(d, counter):
    rows = list(csv.DictReader(open(glob.glob(f"{d}/*counter_collection.csv")[0])))
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", ""))[:52]
        tot[n] += float(r["Counter_Value"]) * 1024.0
        cnt[n] += 1
    return tot, cnt


fetch, cnt = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
steps = max(1, sum(v for k, v in cnt.items() if "stem_fwd_kernel" in k or "nchw_to_nhwc" in k))
tf, tw = 2.0 * sum(fetch.values()), sum(write.values())
print(f"steps profiled: {steps};  HBM traffic per step: read {tf / steps / 1e9:.1f} GB (FETCH_SIZE x2)  write {tw / steps / 1e9:.1f} GB  total {(tf + tw) / steps / 1e9:.1f} GB")
print(f"{'kernel':54s} {'launches/step':>13s} {'read MB/launch':>15s} {'write MB/launch':>16s} {'GB/step':>8s}")
for k in sorted(cnt, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0)))[:28]:
    n = cnt[k]
    print(f"{k:54s} {n / steps:13.1f} {2 * fetch[k] / n / 1e6:15.2f} {write.get(k, 0.0) / n / 1e6:16.2f} {(2 * fetch[k] + write.get(k, 0.0)) / steps / 1e9:8.2f}")
