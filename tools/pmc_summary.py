#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (name prefix) the mean counter value per launch."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "encode_frames" in k or "decode_frames" in k or "compact" in k:
                print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
