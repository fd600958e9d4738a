"""Summarises rocprofv3 --pmc csv output: mean counter value per kernel name per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")[:60]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in acc.items():
    if "gmm_score" not in name and "nnet" not in name and "gemm" not in name:
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print("   %-32s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
