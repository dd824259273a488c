"""Prints one steady-state step of a rocprofv3 kernel trace: every kernel with queue, start, duration."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
c1 = [i for (i, r) in enumerate(rows) if 'conv1_kernel' in r['Kernel_Name']]
a, b = c1[-4], c1[-3]
t0 = int(rows[a]['Start_Timestamp'])
lo = t0 - 200000
for r in rows:
    st = int(r['Start_Timestamp'])
    if st < lo or st > int(rows[b]['Start_Timestamp']) + 100000:
        continue
    n = r['Kernel_Name']
    short = n.split('(')[0].replace('(anonymous namespace)::', '').replace('void ', '')[-46:]
    print('q%-2s %9.1f %8.1f  %s' % (r['Queue_Id'], (st - t0)/1e3, (int(r['End_Timestamp']) - st)/1e3, short))
