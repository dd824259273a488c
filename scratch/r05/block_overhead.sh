#!/bin/bash
# r05: what the driver's 20-step blocks pay per block (fill + drain) under the worker's long sleep, the poll interval, batches in flight
run() { env $1 python bench.py --steps $2 --warmup 10 --no-cpu-baseline --no-dropin-surface --no-single-image $3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 steps $2 $3:', d['value'], d['ms_per_step'], d['host_cpu_ms_per_step'])"; }
for rep in 1 2; do
  run "X=1" 100 ""
  run "X=1" 20 ""
  run "EAE_WORKER_LONG_SLEEP=0" 20 ""
  run "EAE_WORKER_LONG_SLEEP=0 EAE_WORKER_SEQUENCE_POLL_SECONDS=0.0002" 20 ""
  run "X=1" 20 "--coder-streams 5"
  run "EAE_WORKER_LONG_SLEEP=0" 20 "--coder-streams 5"
done
for rep in 1 2; do timeout 100 python scratch/r05/single_forms.py 2>&1 | tail -1; done
