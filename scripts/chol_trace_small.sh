#!/bin/bash
# Kernel-trace statistics of gpbo_fit at one size: scripts/chol_trace_small.sh <tag> <n>   (run on the GPU box from the repository root)
cd "$(dirname "$0")/.." || exit 1
R=$PWD
O=$R/gpurun_out/choltrace_$1
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o kt -- python3 "$R/scripts/chol_timing.py" "$2" > "$O/out.txt" 2> "$O/err.txt" || exit 1
cp $(find "$O/kt" -name '*kernel_stats.csv' | head -1) "$O/kernel_stats.csv"
find "$O" -name '*.db' -delete
cut -c1-60,100- "$O/kernel_stats.csv" | head -14
