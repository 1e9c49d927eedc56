"""Per-step timeline of a rocprofv3 kernel trace: how much of each kernel's time runs alone / overlapped with another stream's kernels.
usage: python tools/timeline.py <rocprof dir> [step marker kernel substring, default adamw]"""
import collections
import csv
import glob
import re
import sys


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    marker = sys.argv[2] if len(sys.argv) > 2 else "adamw"
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    marks = [e for e in ev if marker in e[2]]
    t0, t1 = marks[-2][1], marks[-1][1]
    win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    print(f"step window {(t1 - t0) / 1e6:.2f} ms, {len(win)} kernels, sum of kernel times {sum(e[1] - e[0] for e in win) / 1e6:.2f} ms")
    pts = []
    for i, (s, e, n) in enumerate(win):
        pts.append((s, 1, i))
        pts.append((e, -1, i))
    pts.sort()
    active, last = set(), pts[0][0]
    excl, shared, conc = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(float)
    for t, d, i in pts:
        dt = (t - last) / 1e6
        if dt > 0:
            conc[len(active)] += dt
            if len(active) == 1:
                excl[win[next(iter(active))][2]] += dt
            else:
                for a in active:
                    shared[win[a][2]] += dt
        last = t
        if d == 1:
            active.add(i)
        else:
            active.discard(i)
    print("wall time by number of kernels in flight:", {k: round(v, 1) for k, v in sorted(conc.items())})

    def short(n):
        return re.sub(r"\(.*", "", n).replace("void ", "")[:64]

    tot = collections.defaultdict(lambda: [0.0, 0.0])
    for n, v in excl.items():
        tot[short(n)][0] += v
    for n, v in shared.items():
        tot[short(n)][1] += v
    print(f"{'kernel':66s} alone ms  overlapped ms")
    for n, (a, b) in sorted(tot.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:40]:
        print(f"{n:66s} {a:8.2f} {b:8.2f}")


main()
