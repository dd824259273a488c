#!/bin/bash
# Round-6 evidence on the GPU box (bash scratch/r06/collect_final.sh): the driver's command, the default line, kernel trace + stats of the
# launch-by-launch schedule (the source of `roofline`) and of the product mode, and the PMC passes -- counters only, one counter set
# per pass, the program directly behind `--` -- summarised per kernel (profiles/make_pmc_summary.py, profiles/make_traffic.py).
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06_final
mkdir -p $OUT
python bench.py --steps 20 --warmup 10 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_onestream -- python3 $ROOT/bench.py --no-cpu-baseline --no-side --no-dropin-surface --no-transforms-alone --transform-streams 1 --no-graphs --coder-streams 3 --steps 30 --warmup 5 --min-seconds 0 --max-blocks 1 > $OUT/bench_onestream_under_rocprof.json 2> $OUT/trace_onestream.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_product -- python3 $ROOT/bench.py --no-cpu-baseline --no-side --no-dropin-surface --no-transforms-alone --steps 30 --warmup 5 --min-seconds 0 --max-blocks 1 > $OUT/bench_product_under_rocprof.json 2> $OUT/trace_product.err
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
            "insts:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "active:SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
            "wait:SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  name=${pass%%:*}; counters=${pass#*:}
  timeout 400 rocprofv3 --pmc $counters --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-side --no-dropin-surface --no-transforms-alone --transform-streams 1 --no-graphs --coder-streams 3 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pmc pass $name rc=$?"
done
cd $ROOT
python profiles/make_pmc_summary.py $OUT $OUT/pmc_summary.json > $OUT/pmc_summary.txt; tail -12 $OUT/pmc_summary.txt
cp profiles/traffic_conv_gemm.json $OUT/traffic_conv_gemm_before.json
python profiles/make_traffic.py $OUT 24 > $OUT/traffic.log 2>&1; tail -5 $OUT/traffic.log; cp profiles/traffic_conv_gemm.json $OUT/traffic_conv_gemm.json
for t in onestream product; do s=$(find $OUT/trace_$t -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/trace_${t}_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*.db" -delete
du -sh $OUT; ls $OUT
