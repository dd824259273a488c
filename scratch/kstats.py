import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(r['Name'].replace('(anonymous namespace)::', '')[:70].ljust(70), r['Calls'].rjust(4), round(float(r['AverageNs'])/1e3, 1))
