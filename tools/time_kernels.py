"""Per-kernel median durations of one encoder pass, from a rocprofv3 kernel trace (GPU box):
    rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 tools/run_scattn_once.py 18
    python tools/time_kernels.py DIR/run_kernel_trace.csv"""
import csv
import statistics
import sys
from collections import defaultdict

d = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "gmf::" in k:
        big = [x for x in v if x > 0.5 * max(v)]
        print(f"{k[:60]:60s} n={len(v):4d} median(big)={statistics.median(big):9.1f} us  total={sum(v)/1e3:8.2f} ms")
