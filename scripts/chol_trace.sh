#!/bin/bash
# Per-launch durations of the factorisation's kernels at one size (kernel trace of scripts/chol_timing.py <n>):
#   scripts/chol_trace.sh <n>     -> gpurun_out/chol_trace_<n>.csv (Kernel_Name, grid, start, end of every launch)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
R=$PWD
N=${1:-32768}
O=$R/gpurun_out/chol_trace_$N
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$O/kt" -o kt -- python3 "$R/scripts/chol_timing.py" $N > "$O/out.txt" 2> "$O/kt.err" || exit 1
cd "$R" || exit 1
python3 - "$O" <<'PY'
import csv, sys, glob
o = sys.argv[1]
f = glob.glob(o + '/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(o + '.csv', 'w') as w:
    w.write('kernel,grid_x,grid_y,start_ns,dur_ns\n')
    for r in rows:
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('gpbo::', '').replace(', ', ' ')
        w.write('%s,%s,%s,%s,%d\n' % (name, r['Grid_Size_X'], r['Grid_Size_Y'], r['Start_Timestamp'], int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
print(len(rows), 'launches')
PY
rm -rf "$O/kt"
cat "$O/out.txt"
