#!/bin/bash
# round 6: (1) FP64 against integer floor(p * range) in the encoder core's chain; (2) conv3 in the launch-by-launch leg with and without the coder
# beside it; (3) the bench with its child legs in front
cd "$GRAFT_REPO_ROOT"
./scratch/r06/probe_floor > gpurun_out/r06_probe_floor.log 2>&1
EAE_BENCH_NO_CODER=1 python bench.py --no-side --no-cpu-baseline --no-dropin-surface 2>/dev/null | tail -1 > gpurun_out/r06_roof_nocoder.json
python bench.py --no-side --no-cpu-baseline --no-dropin-surface 2>/dev/null | tail -1 > gpurun_out/r06_roof_coder.json
python bench.py > gpurun_out/r06_bench_b.json 2> gpurun_out/r06_bench_b.err
cat gpurun_out/r06_probe_floor.log
python - <<'PY'
import json
for name in ('nocoder', 'coder'):
    d = json.loads(open('gpurun_out/r06_roof_%s.json' % name).read())
    print(name, d['value'], d['roofline']['per_launch_ms'], {k: v['avg_ms'] for (k, v) in d['roofline']['per_kernel'].items()})
d = json.loads(open('gpurun_out/r06_bench_b.json').read().strip().splitlines()[-1])
print(d['value'], d['library_user'], d['single_image'], d['cpu_baseline']['value'], d['cpu_baseline']['repetitions'], d['cpu_baseline']['spread'], d['dropin_surface']['code_lossless']['value'])
PY
