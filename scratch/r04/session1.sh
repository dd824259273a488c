#!/bin/bash
# r04 GPU session 1: RCCL with one rank; the decoder-core hunt (variants x neighbours); the guard test on the failing form
mkdir -p gpurun_out/r04
L=gpurun_out/r04
timeout 600 python -m pytest tests/test_bench_launcher.py -m gpu -x -q -k rccl > $L/s1_rccl.log 2>&1; echo "rccl rc $?" >> $L/s1_rccl.log
for v in topup_p0 topup_p3 ship_p0 ship_p3; do
  EAE_HIP_LIB=scratch/r04/libs/$v/libeae_hip.so timeout 400 python scratch/r04/decode_hunt.py 1.0 6 2>&1 | grep -v amdgpu.ids >> $L/s1_hunt.log
done
EAE_HIP_LIB=scratch/r04/libs/topup_p0/libeae_hip.so timeout 400 python scratch/r04/decode_hunt.py 0.125 6 2>&1 | grep -v amdgpu.ids >> $L/s1_hunt.log
for v in topup_p3 topup_p0; do
  echo "=== guard test on $v" >> $L/s1_guard.log
  EAE_HIP_LIB=scratch/r04/libs/$v/libeae_hip.so timeout 600 python -m pytest tests/test_coder_device.py -q -k next_to_mfma 2>&1 | tail -15 >> $L/s1_guard.log
done
echo "=== guard test on the shipped library" >> $L/s1_guard.log
timeout 600 python -m pytest tests/test_coder_device.py -q -k next_to_mfma 2>&1 | tail -3 >> $L/s1_guard.log
tail -5 $L/s1_rccl.log; cat $L/s1_hunt.log; cat $L/s1_guard.log
