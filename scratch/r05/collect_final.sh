#!/bin/bash
# Round-5 evidence on the GPU box: the driver's command, a kernel trace + stats of a short run of the same command, PMC passes.
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05_final
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 10 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
cd /tmp && export TMPDIR=/tmp
# one-stream schedule (the source of `roofline`): every kernel's average next to the HIP-event figures of the line
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_onestream -- python3 $ROOT/bench.py --no-cpu-baseline --no-side --no-dropin-surface --transform-streams 1 --no-graphs --coder-streams 3 --steps 30 --warmup 5 --min-seconds 0 --max-blocks 1 > $OUT/bench_onestream_under_rocprof.json 2> $OUT/trace_onestream.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_product -- python3 $ROOT/bench.py --no-cpu-baseline --no-side --no-dropin-surface --steps 30 --warmup 5 --min-seconds 0 --max-blocks 1 > $OUT/bench_product_under_rocprof.json 2> $OUT/trace_product.err
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; counters=${pass#*:}
  timeout 400 rocprofv3 --pmc $counters --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-side --no-dropin-surface --transform-streams 1 --no-graphs > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
cd $ROOT
for t in onestream product; do s=$(find $OUT/trace_$t -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/trace_${t}_kernel_stats.csv; done
find $OUT -name "*kernel_trace.csv" -delete
du -sh $OUT; ls $OUT
