#!/bin/bash
# Kernel-trace statistics (and, with PMC=1, the three counter passes) of one bench.py workload; run on the GPU box from
# the repository root:   scripts/profile_bench.sh <tag> [bench.py arguments]   (cfg2 alone: --no-cfg3 --no-cfg45)
# Summaries land in gpurun_out/prof_<tag>/ (copy what is to be kept into profiles/).
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
R=$PWD
TAG=$1; shift
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp || exit 1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o kt -- python3 "$R/bench.py" --no-cpu-baseline "$@" > "$O/bench_under_trace.json" 2> "$O/kt.err" || exit 1
cp $(find "$O/kt" -name '*kernel_stats.csv' | head -1) "$O/kernel_stats.csv"
if [ "${PMC:-0}" = "1" ]; then
  timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$O/fetch" -o f -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> "$O/f.err" || exit 1
  timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$O/write" -o w -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> "$O/w.err" || exit 1
  timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d "$O/sq" -o s -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> "$O/s.err" || exit 1
  # L2 hit rate (TCC_HIT / (TCC_HIT + TCC_MISS)): tells how much of the operand re-reading stops at the XCD's L2; what
  # misses shows up in FETCH_SIZE (L2 fills: Infinity Cache hits and HBM reads together -- no counter separates the two)
  timeout -s KILL 200 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d "$O/l2" -o l -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> "$O/l.err" || echo "L2 pass failed (see l.err)"
  cd "$R" || exit 1
  python3 scripts/summarize_pmc.py $(find "$O/fetch" "$O/write" "$O/sq" "$O/l2" -name '*counter_collection.csv' | sort) > "$O/pmc_summary.txt"
fi
find "$O" -name '*.db' -delete; find "$O" -name '*counter_collection.csv' -size +20M -delete; find "$O/kt" -name '*kernel_trace.csv' -size +20M -delete
head -8 "$O/kernel_stats.csv"
[ -f "$O/pmc_summary.txt" ] && head -12 "$O/pmc_summary.txt"
exit 0
