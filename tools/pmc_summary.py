"""Average rocprofv3 --pmc counters per launch of the kernels whose name contains a pattern.
usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> <kernel-name substring>"""
import csv, glob, os, sys
from collections import defaultdict
root, pat = sys.argv[1], sys.argv[2]
acc, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[row["Counter_Name"]] += 1
for k in sorted(acc):
    print("%-24s %16.1f  (avg of %d launches)" % (k, acc[k] / cnt[k], cnt[k]))
