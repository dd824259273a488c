#!/bin/bash
mkdir -p gpurun_out/r04
L=gpurun_out/r04
timeout 300 ./scratch/r04/libs/probe_last_vgpr 48 20000 3 > $L/s14_probe_last.log 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $L/s14_gpu_tests.log
timeout 900 python bench.py > $L/s14_bench.json 2> $L/s14_bench.err
grep -i "mfma" $L/s14_probe_last.log; cat $L/s14_gpu_tests.log; python scratch/r03_line.py $L/s14_bench.json 2>/dev/null || head -c 1500 $L/s14_bench.json
