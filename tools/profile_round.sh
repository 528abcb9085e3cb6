#!/bin/bash
# The round's committed profiles, one GPU job:  bash tools/profile_round.sh r05      (results under gpurun_out/, copied to profiles/ by hand)
R=${1:-r05}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/${R}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_stats -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sweep > gpurun_out/${R}_stats.log 2>&1
f=$(find gpurun_out/${R}_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${R}_bench_n1_kernel_stats.csv
echo "stats done"; head -8 gpurun_out/${R}_bench_n1_kernel_stats.csv
bash tools/pmc_scattn.sh 18 ${R}_pmc_scattn k_scattn_h2p > gpurun_out/${R}_scattn_h2p_pmc.json 2> gpurun_out/${R}_pmc_scattn.err
echo "pmc scattn done"
bash tools/pmc_scattn.sh 18 ${R}_pmc_linear k_linear_h2 > gpurun_out/${R}_linear_h2_pmc.json 2> gpurun_out/${R}_pmc_linear.err
echo "pmc linear done"
python3 - <<PY
import json
for k in ("scattn_h2p", "linear_h2"):
    d = json.load(open("gpurun_out/${R}_%s_pmc.json" % k))
    print(k, d["csrc_sha16"], d["derived"], d["avg_launch_ms_under_pmc"])
PY
