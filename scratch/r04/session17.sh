#!/bin/bash
mkdir -p gpurun_out/r04
L=gpurun_out/r04
ROOT=$(pwd)
timeout -k 10 2400 python -m pytest tests -m gpu -x -q < /dev/null 2>&1 | tail -8 > $L/s17_gpu_tests.log
cat $L/s17_gpu_tests.log
timeout -k 10 900 python bench.py < /dev/null > $L/s17_bench.json 2> $L/s17_bench.err
python scratch/r03_line.py r04-s17 < $L/s17_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $ROOT/$L/s17_pmc_active -- python3 $ROOT/bench.py --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-side --transform-streams 1 --no-graphs --coder-streams 3 < /dev/null > $ROOT/$L/s17_pmc_active.json 2> $ROOT/$L/s17_pmc_active.err
cd $ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r04/s17_pmc_active/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for path in f:
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'][:60]
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
        if row['Counter_Name'] == 'SQ_LDS_BANK_CONFLICT': cnt[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_LDS_BANK_CONFLICT', 0))[:8]:
    n = max(cnt[k], 1)
    print('%-60s launches %4d  per launch: LDS bank conflict %12.0f  active LDS %12.0f  active any %14.0f' % (k, n, v.get('SQ_LDS_BANK_CONFLICT', 0)/n, v.get('SQ_ACTIVE_INST_LDS', 0)/n, v.get('SQ_ACTIVE_INST_ANY', 0)/n))
PY
find $L -name "*counter_collection.csv" -delete
