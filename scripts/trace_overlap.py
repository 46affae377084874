"""From a rocprofv3 kernel trace: queues used, and how many kernels run at once (time-weighted) over the last third."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[2 * len(rows) // 3:]
print('kernels', len(rows), 'queues', collections.Counter(r['Queue_Id'] for r in rows))
ev = []
for r in rows:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
cur = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print('concurrency histogram (fraction of time):', {k: round(v / tot, 3) for k, v in sorted(hist.items())})
dur = collections.defaultdict(list)
for r in rows:
    dur[r['Kernel_Name'].split('(')[0][-30:]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print('%-32s n=%4d avg %.1f us total %.2f ms' % (k, len(v), sum(v) / len(v), sum(v) / 1e3))
print('wall %.2f ms' % ((int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6))
