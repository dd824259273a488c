#!/bin/bash
# result worker: events (round 4) against the device-written step counter; headline, host CPU per step, one image at a time
mkdir -p gpurun_out/r05f
for mode in events sequence; do
  for poll in 0.00005 0.0002; do
    [ "$mode" = events ] && [ "$poll" = 0.00005 ] && continue
    EAE_WORKER_WAIT=$mode EAE_WORKER_SEQUENCE_POLL_SECONDS=$poll python bench.py --no-cpu-baseline --no-dropin-surface --steps 60 --warmup 10 > gpurun_out/r05f/bench_${mode}_${poll}.json 2> gpurun_out/r05f/bench_${mode}_${poll}.err
    python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05f/bench_${mode}_${poll}.json') if l.startswith('{')][-1])
print('$mode', '$poll', 'value', d['value'], 'ms/step', d['ms_per_step'], 'host cpu', d['host_cpu_ms_per_step'], 'one-stream cpu', d['one_stream_leg']['host_cpu_ms_per_step'],
      'single', d['single_image']['latency_ms'], d['single_image']['ms_per_image'], 'other', [(o['workload'], o['value']) for o in d['other_shapes']], 'pcie', d['pcie_inclusive']['value'])
PY
  done
done
