#!/bin/bash
# Counter passes over the factorisation at one size (scripts/chol_timing.py <n>): L2 fills and the MFMA / wait shares.
#   scripts/chol_pmc.sh <n>   -> gpurun_out/chol_pmc_<n>.txt
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
R=$PWD
N=${1:-32768}
O=$R/gpurun_out/chol_pmc_$N
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp || exit 1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$O/fetch" -o f -- python3 "$R/scripts/chol_timing.py" $N > /dev/null 2> "$O/f.err" || exit 1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d "$O/sq" -o s -- python3 "$R/scripts/chol_timing.py" $N > /dev/null 2> "$O/s.err" || exit 1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d "$O/tcc" -o t -- python3 "$R/scripts/chol_timing.py" $N > /dev/null 2> "$O/t.err" || exit 1
cd "$R" || exit 1
python3 scripts/summarize_pmc.py $(find "$O/fetch" "$O/sq" "$O/tcc" -name '*counter_collection.csv' | sort) > "$O.txt"
rm -rf "$O/fetch" "$O/sq" "$O/tcc"
grep -E "syrk_update<true|inv_step" "$O.txt"
