"""Summary of a rocprofv3 kernel trace (kernel_trace.csv): over the middle half of the traced span, the share of time with at least
one kernel running, the mean number running, per queue busy shares, and per kernel name count / mean duration."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-60:], r.get('Queue_Id', '?')) for r in rows]
ev.sort()
(t0, t1) = (ev[0][0], max(e[1] for e in ev))
(a, b) = (t0 + (t1 - t0)*0.4, t0 + (t1 - t0)*0.9)
mid = [e for e in ev if e[0] >= a and e[1] <= b]
points = sorted([(e[0], 1) for e in mid] + [(e[1], -1) for e in mid])
(busy, weighted, running, last) = (0, 0, 0, a)
for (t, d) in points:
    if running > 0:
        busy += t - last
        weighted += (t - last)*running
    running += d
    last = t
span = b - a
print('kernels in the window %d over %.3f ms: some kernel running %.1f %% of the time, mean number running %.2f' % (len(mid), span/1e6, 100.*busy/span, weighted/span))
per_q = collections.defaultdict(int)
for e in mid:
    per_q[e[3]] += e[1] - e[0]
print('busy share per queue:', {q: round(v/span, 3) for (q, v) in sorted(per_q.items())})
per_k = collections.defaultdict(list)
for e in mid:
    per_k[e[2]].append(e[1] - e[0])
for (k, v) in sorted(per_k.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print('  %-62s n %5d mean %8.1f us total share %.3f' % (k, len(v), sum(v)/len(v)/1e3, sum(v)/span))
