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
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16   # noqa: E402  (ties the summary to the source tree it was taken on; bench.py checks it)
c = {k: sums[k] / cnts[k] for k in sums}
derived = {}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> doubled;
    # WRITE_SIZE exact; both in KB
    derived["traffic_bytes_per_launch"] = int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
    derived["mfma_pipe_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
if "GRBM_GUI_ACTIVE" in c and nd:
    derived["clock_GHz"] = (c["GRBM_GUI_ACTIVE"] / 8.0) / (dur / nd)
if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
    derived["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
wl = {"pairs": int(os.environ.get("PMC_PAIRS", "32")), "n_corr": int(os.environ.get("PMC_NCORR", "5000")), "tokens": 196}
out = {"csrc_sha16": csrc_sha16(), "workload": wl, "derived": derived, "kernel": kname, "launches_per_counter": dict(cnts), "counters": {c: sums[c] / cnts[c] for c in sums},
       "avg_launch_ms_under_pmc": dur / max(nd, 1) / 1e6}
print(json.dumps(out, indent=1))
