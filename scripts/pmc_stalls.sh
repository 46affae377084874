#!/bin/bash
# Two counter passes that show what the sweep kernel's waves wait for (LDS conflicts, FIFO back-pressure, instruction mix).
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out/pmc_stalls; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp || exit 1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL -d "$O/a" -o a -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline ${BENCH_ARGS} > /dev/null 2> "$O/a.err" || exit 1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES -d "$O/b" -o b -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline ${BENCH_ARGS} > /dev/null 2> "$O/b.err" || exit 1
cd "$R" || exit 1
python3 scripts/summarize_pmc.py $(find "$O/a" "$O/b" -name '*counter_collection.csv' | sort) | grep -E "k_trmm_sumsq|^#" > "$O/summary.txt"
find "$O" -name '*counter_collection.csv' -delete
cat "$O/summary.txt"
