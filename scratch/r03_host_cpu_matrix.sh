#!/bin/bash
# Which runtime setting / codec mode the rank's busy runtime thread (3 ms of host CPU per 3 ms step, mostly system time) follows.
OUT=gpurun_out/${1:-r03_k}; mkdir -p $OUT
run() { echo "== $*"; env "$@" timeout 200 python scratch/r03_host_cpu.py $STEPS $GRAPHS $TS 2>&1 | grep -v amdgpu.ids; }
STEPS=600; GRAPHS=1; TS=2
{
run A=baseline
run EAE_WORKER_POLL_SECONDS=0
run EAE_WORKER_POLL_SECONDS=0.001
run HSA_ENABLE_INTERRUPT=0
run AMD_DIRECT_DISPATCH=0
run ROC_CPU_WAIT_FOR_SIGNAL=0
run GPU_MAX_HW_QUEUES=8
run ROC_ACTIVE_WAIT_TIMEOUT=0
GRAPHS=0; TS=2; run A=launches_two_streams
GRAPHS=0; TS=1; run A=launches_one_stream
} | tee $OUT/host_cpu_matrix.txt
