import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'conv_gemm_wave' in r['Kernel_Name'] or 'gdn_kernel' in r['Kernel_Name']]
t0 = int(rows[0]['Start_Timestamp'])
# print launches 60..80
for r in rows[int(sys.argv[2]):int(sys.argv[2]) + int(sys.argv[3])]:
    st = int(r['Start_Timestamp']); en = int(r['End_Timestamp'])
    print('%10.1f %8.1f grid %s %s' % ((st - t0)/1e3, (en - st)/1e3, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r['Kernel_Name'][40:75]))
