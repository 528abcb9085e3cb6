"""Aggregate rocprofv3 --pmc passes: mean counter value per launch of the kernels whose name contains PATTERN.
    python tools/pmc_summary.py 'gpurun_out/pmc_g*' k_scattn"""
import csv
import glob
import json
import sys
from collections import defaultdict

dirs, pat = sys.argv[1], sys.argv[2]
sums, cnts = defaultdict(float), defaultdict(int)
dur, nd = 0.0, 0
kname = None
for d in sorted(glob.glob(dirs)):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = defaultdict(float)
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                kname = row["Kernel_Name"]
                per_dispatch[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (_, c), v in per_dispatch.items():
            sums[c] += v
            cnts[c] += 1
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                dur += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                nd += 1
out = {"kernel": kname, "launches_per_counter": dict(cnts), "counters": {c: sums[c] / cnts[c] for c in sums},
       "avg_launch_ms_under_pmc": dur / max(nd, 1) / 1e6}
print(json.dumps(out, indent=1))
