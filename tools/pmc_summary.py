#!/usr/bin/env python3
"""Summarise rocprofv3 PMC runs into per-kernel HBM traffic.

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section),
on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes, i.e. reads exactly HALF of the bytes of a
coalesced streaming read: it is doubled here.  WRITE_SIZE is exact for streaming stores.  The two
counters need separate passes (TCC slot budget), hence two input files.
"""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, [])
        w = write.get(k, [])
        rd = 2.0 * 1024.0 * sum(f) / len(f) if f else None
        wr = 1024.0 * sum(w) / len(w) if w else None
        out[k] = dict(launches=max(len(f), len(w)), read_bytes_per_launch=rd, write_bytes_per_launch=wr,
                      hbm_bytes_per_launch=(rd or 0) + (wr or 0),
                      note="FETCH_SIZE x2 (gfx950 correction) x1024; WRITE_SIZE x1024")
    rows = dict(out)
    # what these figures are valid for: the exact kernel sources (bench.py reports `traffic_stale` when they differ)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from indigo_amd.build import source_hash
    out["_meta"] = dict(csrc_sha16=source_hash(), kernels=sorted(rows))
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in rows.items():
        print("%-60s %4d launches  read %8.3f GB  write %8.3f GB" % (
            k[:60], v["launches"], (v["read_bytes_per_launch"] or 0) / 1e9, (v["write_bytes_per_launch"] or 0) / 1e9))


if __name__ == "__main__":
    main()
