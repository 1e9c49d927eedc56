#!/usr/bin/env python
"""Per-kernel-family averages of derived utilisation counters from rocprofv3 --pmc runs of the bench command.

    python tools/pmc_util.py <dir with *counter_collection.csv> [<dir> ...]

Each directory holds one pass (`rocprofv3 --kernel-trace --pmc VALUBusy MfmaUtil ...`); the table lists, per kernel family (template
arguments stripped), the number of dispatches seen and the plain mean of every counter over them."""
import collections
import csv
import glob
import re
import sys


def family(name):
    n = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    return re.sub(r"<.*", "", n)


acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[family(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
names = sorted({c for fam in acc.values() for c in fam})
print(f"{'kernel family':34s} {'dispatches':>10s} " + " ".join(f"{c:>20s}" for c in names))
for fam, cs in sorted(acc.items(), key=lambda kv: -max(v[1] for v in kv[1].values())):
    n = max(v[1] for v in cs.values())
    if n < 2:
        continue
    print(f"{fam[:34]:34s} {n:10d} " + " ".join(f"{(cs[c][0] / cs[c][1]) if c in cs else float('nan'):20.2f}" for c in names))
