#!/bin/bash
mkdir -p gpurun_out/r04
L=gpurun_out/r04
timeout -k 10 1500 python -m pytest tests -m gpu -x -q -s -k "mid or order_sensitivity or outside_the_mid" < /dev/null > $L/s15_new_tests.log 2>&1
tail -12 $L/s15_new_tests.log
timeout -k 10 2400 python -m pytest tests -m gpu -x -q < /dev/null 2>&1 | tail -8 > $L/s15_gpu_tests.log
cat $L/s15_gpu_tests.log
timeout -k 10 900 python bench.py < /dev/null > $L/s15_bench.json 2> $L/s15_bench.err
python scratch/r03_line.py r04-mid < $L/s15_bench.json
