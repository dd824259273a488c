#!/bin/bash
# kernel trace of the pipelined one-image leg: how many kernels run at once, how much of the time nothing runs
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04/${1:-s60}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
H=${H:-512} W=${W:-768} timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/scratch/r04/worker_profile.py < /dev/null > $OUT/run.log 2> $OUT/err.txt
cd $ROOT
grep "ms per image" $OUT/run.log
python3 - $OUT/trace <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
# the last 60 % of the trace = the 2,000 timed steps in their steady state
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0])*0.5
t_hi = rows[0][0] + (rows[-1][1] - rows[0][0])*0.95
ev = []
n = 0
for (s, e, k) in rows:
    if s >= t_lo and e <= t_hi:
        ev.append((s, 1)); ev.append((e, -1)); n += 1
ev.sort()
cur = 0; last = ev[0][0]; hist = {}
for (t, d) in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += d
tot = sum(hist.values())
print('kernels in the window: %d over %.1f ms = %.1f us per kernel launch; time share by number of kernels running at once:' % (n, tot/1e6, tot/1e3/n))
print('  ' + '  '.join('%d: %.1f %%' % (k, 100.*v/tot) for (k, v) in sorted(hist.items())))
conv1 = [s for (s, e, k) in rows if 'conv1_kernel' in k and s >= t_lo and s <= t_hi]
print('  conv1 to conv1: %.1f us' % ((conv1[-1] - conv1[0])/1e3/(len(conv1) - 1)))
PY
find $OUT/trace -name "*kernel_trace.csv" -delete
