#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from counter_collection.csv files.
usage: pmc_summary.py <dir-or-csv> [...]  (prints one line per input and kernel)"""
import collections
import csv
import glob
import os
import sys

for arg in sys.argv[1:]:
    files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "synth" in k or "build_tables" in k:
                continue
            # the pass (its directory under the argument), never an absolute path of the box
            print(os.path.relpath(os.path.dirname(f), arg).split(os.sep)[0], "|", k)
            for c, v in sorted(cs.items()):
                print(f"    {c:34s} {sum(v)/len(v):16.1f}  (n={len(v)})")
