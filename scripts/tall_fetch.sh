#!/bin/bash
# L2 fills of the variance product with the 256 x 128 tile (cfg3, one step): for the record of the experiment.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out/prof_tall; mkdir -p $O
export TMPDIR=/tmp GPBO_TRMM_TALL=1
cd /tmp || exit 1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$O/fetch" -o f -- python3 "$R/bench.py" --config cfg3 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$O/f.err" || exit 1
cd "$R" && python3 scripts/summarize_pmc.py $(find "$O/fetch" -name '*counter_collection.csv') > "$O/pmc_summary.txt"
find "$O" -name '*.db' -delete; find "$O" -name '*counter_collection.csv' -size +20M -delete
grep -i "trmm" "$O/pmc_summary.txt" | head -5
