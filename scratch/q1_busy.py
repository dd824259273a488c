"""Per-step busy / idle time of the transform queue from a rocprofv3 kernel trace."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
main_q = collections.Counter(r['Queue_Id'] for r in rows).most_common(1)[0][0]
q1 = [r for r in rows if r['Queue_Id'] == main_q]
starts = [i for (i, r) in enumerate(q1) if 'conv1_kernel' in r['Kernel_Name']]
starts = starts[-8:]
per = collections.defaultdict(list)
for a, b in zip(starts[:-1], starts[1:]):
    period = (int(q1[b]['Start_Timestamp']) - int(q1[a]['Start_Timestamp']))/1e3
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in q1[a:b])/1e3
    per['period'].append(period); per['busy'].append(busy)
    for r in q1[a:b]:
        n = r['Kernel_Name']
        key = 'conv1' if 'conv1_kernel' in n else 'tconv3' if 'tconv3' in n else ('gemm_%s' % r['Grid_Size_X'] if 'conv_gemm' in n else 'gdn' if 'gdn' in n else 'quant' if 'quantize' in n else 'small')
        per[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3)
n = len(per['period'])
print('steps', n, {k: round(sum(v)/n, 1) for (k, v) in per.items()})
others = [r for r in rows if r['Queue_Id'] != main_q and 'coder' in r['Kernel_Name'] or 'decoder_maps' in r['Kernel_Name']]
if others:
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3 for r in others[-12:]]
    print('coder kernels (us):', [round(x) for x in d])
