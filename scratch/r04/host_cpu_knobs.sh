#!/bin/bash
# r04: does any runtime setting lower the host CPU a rank spends per step (3.8 ms of process CPU per 3.0 ms step, 3.1 of it in the HIP
# runtime's event thread: DESIGN.md section 7)? The product mode, no side legs; one line per setting.
mkdir -p gpurun_out/r04
L=gpurun_out/r04/s18_host_cpu_knobs.log; : > $L
run() {
  name="$1"; shift
  out=$(env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-side --steps 60 --min-seconds 0.5 < /dev/null 2>/dev/null | tail -1)
  python - "$name" "$out" >> $L <<'PY'
import json, sys
name, raw = sys.argv[1], sys.argv[2]
try:
    d = json.loads(raw)
    print('%-44s %8.1f Mpx/s  %6.3f ms/step  host CPU per step %s ms  one-stream leg %s ms' % (name, d['value'], d['ms_per_step'], d['host_cpu_ms_per_step'], d['one_stream_leg']['host_cpu_ms_per_step']))
except Exception as e:
    print('%-44s failed: %s %s' % (name, e, raw[:200]))
PY
}
run "default" A=1
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "DEBUG_HIP_GRAPH_BATCH_SIZE=64" DEBUG_HIP_GRAPH_BATCH_SIZE=64
run "DEBUG_CLR_MAX_BATCH_SIZE=64" DEBUG_CLR_MAX_BATCH_SIZE=64
run "DEBUG_CLR_BATCH_CPU_SYNC_SIZE=64" DEBUG_CLR_BATCH_CPU_SYNC_SIZE=64
run "GPU_STREAMOPS_CP_WAIT=1" GPU_STREAMOPS_CP_WAIT=1
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run "DEBUG_HIP_BLOCK_SYNC=50" DEBUG_HIP_BLOCK_SYNC=50
run "packet capture + dev kernarg" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 HIP_FORCE_DEV_KERNARG=1
cat $L
