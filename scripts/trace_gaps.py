#!/usr/bin/env python3
"""Print the tail of a rocprofv3 kernel trace (csv) as a sequence: kernel, duration, idle gap before it."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
prev = None
for r in rows[-n:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-44s dur %9.1f us  gap %8.1f us' % (r['Kernel_Name'].replace('void gpbo::', '').replace('gpbo::', '')[:44], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
    prev = e
