#!/usr/bin/env python
"""Per-step kernel table of a `tools/bench_models.py --models M` run from its rocprofv3 outputs: the kernel trace (one step = the launches between
two adamw_multi_kernel launches; the last complete step is shown) and, when given, the two PMC passes FETCH_SIZE / WRITE_SIZE (collected
separately; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, as tools/pmc_summary.py does).

    python tools/model_prof_summary.py MODEL IMAGES_PER_STEP prof_dir [pmc_fetch_dir pmc_write_dir]

Prints what is needed to recompute the roofline fraction of the model's line in profiles/r06_bench_models.jsonl: step time, images per
second, algorithmic GFLOP and MB per image, achieved TFLOP/s and TB/s, measured HBM bytes per step."""
import collections
import csv
import glob
import re
import sys

ALGO = {"vit_base": (106.25, 190.9, "mfma"), "mobilevitv2": (24.46, 362.6, "hbm"), "clip": (124.0, None, "mfma")}
HBM_PEAK, MFMA_PEAK = 8.0e12, 2.5e15


def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    return re.sub(r"[<(].*", "", n)[:44]


def main():
    model, imgs, prof = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rows = list(csv.DictReader(open(glob.glob(f"{prof}/*kernel_trace.csv")[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "adamw_multi_kernel" in r["Kernel_Name"]]
    step = rows[ends[-2] + 1: ends[-1] + 1]
    t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
    ms = (t1 - t0) / 1e6
    gf, mb, bound = ALGO.get(model, (None, None, "hbm"))
    print(f"{model}: {imgs} images per step, {len(step)} kernel launches, window {ms:.3f} ms, kernel time {busy / 1e6:.3f} ms, {imgs / ms * 1e3:.0f} img/s (under the profiler)")
    if gf:
        tf = gf * 1e9 * imgs / (ms * 1e-3)
        print(f"  algorithmic {gf} GFLOP/img -> {tf / 1e12:.1f} TFLOP/s = {tf / MFMA_PEAK:.3f} of the dense bf16 MFMA peak (2.5 PFLOP/s)")
    if mb:
        bw = mb * 1e6 * imgs / (ms * 1e-3)
        print(f"  algorithmic {mb} MB/img -> {bw / 1e12:.2f} TB/s = {bw / HBM_PEAK:.3f} of the HBM peak (8 TB/s)")
    fam = collections.defaultdict(lambda: [0, 0])
    for r in step:
        k = short(r["Kernel_Name"])
        fam[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        fam[k][1] += 1
    traffic = {}
    if len(sys.argv) > 5:
        def load(d, counter):
            tot, cnt = collections.defaultdict(float), collections.Counter()
            for r in csv.DictReader(open(glob.glob(f"{d}/*counter_collection.csv")[0])):
                if r["Counter_Name"] == counter:
                    tot[short(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1024.0
                    cnt[short(r["Kernel_Name"])] += 1
            return tot, cnt
        fetch, cnt = load(sys.argv[4], "FETCH_SIZE")
        write, _ = load(sys.argv[5], "WRITE_SIZE")
        nsteps = max(1, cnt.get("adamw_multi_kernel", 1))
        traffic = {k: (2.0 * fetch[k] + write.get(k, 0.0)) / nsteps for k in fetch}
        tot = sum(traffic.values())
        print(f"  measured HBM traffic per step ({nsteps} steps under the counters): {tot / 1e9:.1f} GB = {tot / (ms * 1e-3) / 1e12:.2f} TB/s over the step window")
    print(f"  {'kernel family':44s} {'launches':>8s} {'ms/step':>9s} {'share':>7s} {'GB/step':>9s} {'TB/s':>6s}")
    for k, (ns, n) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:24]:
        tr = traffic.get(k)
        print(f"  {k:44s} {n:8d} {ns / 1e6:9.3f} {ns / busy * 100:6.1f}% " + (f"{tr / 1e9:9.2f} {tr / ns / 1e3:6.2f}" if tr else ""))


if __name__ == "__main__":
    main()
