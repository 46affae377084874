"""Summarise a rocprofv3 kernel trace of one fit: per-queue busy time of the last Cholesky and its wall time."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
names = [r['Kernel_Name'] for r in rows]
zu = [i for i, n in enumerate(names) if 'k_zero_upper' in n]
end = zu[-2] if len(zu) >= 2 else zu[-1]
start = max(i for i in range(end) if 'k_kernel_matrix' in names[i])
sel = rows[start + 1:end]
t0 = min(int(r['Start_Timestamp']) for r in sel)
t1 = max(int(r['End_Timestamp']) for r in sel)
print('wall %.2f ms, %d kernels' % ((t1 - t0) / 1e6, len(sel)))
busy = collections.defaultdict(float)
cnt = collections.Counter()
for r in sel:
    n = r['Kernel_Name'].split('(')[0].replace('gpbo::', '').replace('void ', '')
    key = (r['Queue_Id'], n)
    busy[key] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    cnt[key] += 1
for k in sorted(busy):
    print('queue %s %-26s %5d launches %8.2f ms' % (k[0], k[1], cnt[k], busy[k]))
if len(sys.argv) > 2:
    lo, hi = int(sys.argv[2]), int(sys.argv[3])
    for r in sel[lo:hi]:
        n = r['Kernel_Name'].split('(')[0].replace('gpbo::', '').replace('void ', '')
        print('%-22s q=%s s=%9.1f e=%9.1f dur=%7.1f gx=%s' % (n[:22], r['Queue_Id'], (int(r['Start_Timestamp']) - t0) / 1e3,
              (int(r['End_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Grid_Size_X']))
