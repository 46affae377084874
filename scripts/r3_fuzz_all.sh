#!/bin/bash
# Every randomised differential run against the current binary, one after the other (run on the GPU box from the repository
# root); outputs under gpurun_out/r3_fuzz_*.txt.  Stops at the first run that fails or times out.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out
mkdir -p $O
run() { local name=$1; shift; echo "# $*" > $O/r3_fuzz_$name.txt; timeout -k 10 "$T" python3 -u "$@" >> $O/r3_fuzz_$name.txt 2>&1 || { echo "FAILED: $name"; tail -5 $O/r3_fuzz_$name.txt; exit 1; }; tail -1 $O/r3_fuzz_$name.txt; }
T=200 run multi scripts/fuzz_multi.py 150 7
T=300 run lattice scripts/fuzz_lattice.py 120 3
T=200 run vs_oracle scripts/fuzz_vs_oracle.py 120 3
T=300 run more_vs_oracle scripts/fuzz_more_vs_oracle.py 100 3
T=300 run state_machine scripts/fuzz_state_machine.py 60 40 3
T=200 run backtrack scripts/fuzz_backtrack.py 150 3
