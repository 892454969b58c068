#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, mean counter value per dispatch."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0][-40:]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, counters in sorted(acc.items()):
    if "orbx" not in name and "probe" not in name and "knn2" not in name:
        continue
    n = max(len(v) for v in counters.values())
    print("%-42s dispatches=%d" % (name, n))
    for c, v in sorted(counters.items()):
        print("    %-28s %16.1f" % (c, sum(v) / len(v)))
