#!/bin/bash
# round 6: (1) one image at a time: results formed by the caller with a bounded spin, against the worker (round 5); (2) where a
# CU-masked stream runs; (3) coder streams on CUs of their own, A/B at three rates
cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=16
L=gpurun_out/r06_latency.log; : > $L
for setting in "EAE_RESULT_BY_CALLER=0" "EAE_RESULT_SPIN_SECONDS=0" "EAE_RESULT_SPIN_SECONDS=0.00015" "EAE_RESULT_SPIN_SECONDS=0.0003" "EAE_RESULT_BY_CALLER=0"; do
  env $setting timeout 300 python scratch/r06/latency.py 2>/dev/null | grep "per image" >> $L
done
env LATENCY_SPLIT=0 timeout 300 python scratch/r06/latency.py 2>/dev/null | grep "per image" >> $L
./scratch/r06/cumask_probe > gpurun_out/r06_cumask_probe.log 2>&1
A=gpurun_out/r06_cumask_ab.log; : > $A
for v in "CODER_CUS=0" "CODER_CUS=16" "CODER_CUS=32" "CODER_CUS=0" "CODER_CUS=16 TRANSFORM=all" "CODER_CUS=32 TRANSFORM=all" "CODER_CUS=64" "CODER_CUS=8"; do
  env $v timeout 400 python scratch/r06/cumask_ab.py 2>gpurun_out/r06_cumask_ab.err | grep "Mpx/s" >> $A
done
cat $L gpurun_out/r06_cumask_probe.log $A; tail -3 gpurun_out/r06_cumask_ab.err
