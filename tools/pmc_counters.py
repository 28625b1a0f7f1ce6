#!/usr/bin/env python3
"""Average every counter of one or more rocprofv3 --pmc csv files per kernel.

    python tools/pmc_counters.py out.json a_counter_collection.csv [b_counter_collection.csv ...]
"""
import collections, csv, json, re, sys

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for k in sorted(out):
    print(k[:70])
    for c, v in sorted(out[k].items()):
        print("    %-34s %16.0f" % (c, v))
