#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_bench.sh:
   python3 scripts/make_traffic_json.py <prof_dir> <workload> <n_train> <candidates_per_launch> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

d, workload, n, cpl, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
res = {}
for sub, counter in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    path = glob.glob(os.path.join(d, sub, '*counter_collection.csv'))[0]
    tot, disp = 0.0, set()
    for r in csv.DictReader(open(path)):
        if 'k_trmm_sumsq' in r['Kernel_Name'] and r['Counter_Name'] == counter:
            tot += float(r['Counter_Value'])
            disp.add(r['Dispatch_Id'])
    res[counter] = (tot / len(disp) * 1024.0, len(disp))          # rocprofv3 reports KiB
l2 = {}
for path in glob.glob(os.path.join(d, 'l2', '*counter_collection.csv')):
    acc = collections.defaultdict(float)
    disp = set()
    for r in csv.DictReader(open(path)):
        if 'k_trmm_sumsq' in r['Kernel_Name'] and r['Counter_Name'] in ('TCC_HIT_sum', 'TCC_MISS_sum'):
            acc[r['Counter_Name']] += float(r['Counter_Value'])
            disp.add(r['Dispatch_Id'])
    if disp:
        l2 = {'l2_hits_per_launch': acc['TCC_HIT_sum'] / len(disp), 'l2_misses_per_launch': acc['TCC_MISS_sum'] / len(disp),
              'l2_hit_rate': acc['TCC_HIT_sum'] / (acc['TCC_HIT_sum'] + acc['TCC_MISS_sum'])}
out_d = {
    'kernel': 'k_trmm_sumsq', 'workload': workload, 'n_train': n, 'candidates_per_launch': cpl,
    'fetch_bytes_per_launch': 2.0 * res['FETCH_SIZE'][0], 'write_bytes_per_launch': res['WRITE_SIZE'][0],
    'launches_counted': res['FETCH_SIZE'][1],
    'algorithmic_bytes_per_launch': 8.0 * n * cpl + 8.0 * n * n / 2 + 8.0 * (n / 128) * cpl,
    'how': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py --steps 1 --warmup 0 (scripts/profile_bench.sh); '
           'rocprofv3 reports KiB; FETCH_SIZE doubled per the gfx950 correction for 16-byte-per-lane streaming reads. These are L2 fills '
           '(the counters sit at the L2/fabric boundary): reads served by the Infinity Cache are included. algorithmic = the Ks panel once '
           '(8 n bytes per candidate) + W once (4 n^2 bytes) + the partial sums written',
}
if l2:
    out_d.update(l2)
    out_d['where_the_rereads_are_served'] = (
        'TCC_HIT / (TCC_HIT + TCC_MISS) of the same kernel (a fourth pass): that share of the operand requests stops at the XCD\'s L2; '
        'the misses x 128 B equal the fetched bytes above, i.e. every miss is one 128-byte fill over the fabric.  The fills are '
        'Infinity-Cache hits and HBM reads together: the eight XCDs request the same Ks and W panels within microseconds of each '
        'other, so up to 7 of 8 fills can be served by the 256 MB memory-side cache, but no counter separates the two.')
json.dump(out_d, open(out, 'w'), indent=1)
print(open(out).read())
