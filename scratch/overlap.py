"""Reads a rocprofv3 kernel trace csv; prints, for a window of steps, every kernel with start/end (us) and queue."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
coder = [i for (i, r) in enumerate(rows) if 'coder_maps' in r['Kernel_Name']]
mid = coder[len(coder)*2//3]
for r in rows[mid - 14: mid + 16]:
    name = r['Kernel_Name'].split('(')[0][-40:]
    print('%10.1f %10.1f %8.1f q%s %s grid=%s' % ((int(r['Start_Timestamp']) - t0)/1e3, (int(r['End_Timestamp']) - t0)/1e3,
          (int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3, r.get('Queue_Id', '?'), name, r.get('Grid_Size_X', r.get('Grid_Size', '?'))))
