#!/bin/bash
# Round 4 evidence session (run from the repo root on the GPU box: bash scratch/r04_final.sh <tag>): the GPU suite, the default
# bench line, the same command under rocprofv3 --kernel-trace --stats, the one-stream launch-by-launch schedule under the same
# (the schedule the roofline leg times), and the PMC passes -- counters only, one counter set per pass, the program directly
# behind `--`. Everything lands in gpurun_out/<tag>/; the summaries are copied into profiles/ afterwards by hand.
TAG=${1:-r04_final}
ROOT=$(pwd)
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 2400 python -m pytest tests -m gpu -x -q < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
  tail -4 $OUT/pytest.log
fi
timeout -k 10 1200 python bench.py < /dev/null > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
grep -v amdgpu.ids $OUT/bench.err | tail -c 600
python scratch/r03_line.py default headline < $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-side < /dev/null > $ROOT/$OUT/bench_under_rocprof.json 2> $ROOT/$OUT/trace.err
echo "trace (default mode) rc=$?"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_onestream -- python3 $ROOT/bench.py --no-cpu-baseline --no-side --transform-streams 1 --no-graphs --coder-streams 3 < /dev/null > $ROOT/$OUT/bench_onestream_under_rocprof.json 2> $ROOT/$OUT/trace_onestream.err
echo "trace (one stream) rc=$?"
for bw in 0.125 0.05; do
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_bw$bw -- python3 $ROOT/bench.py --no-cpu-baseline --no-side --bin-width $bw < /dev/null > $ROOT/$OUT/bench_bw${bw}_under_rocprof.json 2> $ROOT/$OUT/trace_bw$bw.err
  echo "trace (bin width $bw) rc=$?"
done
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
            "insts:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "active:SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
            "wait:SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  name=${pass%%:*}; counters=${pass#*:}
  timeout -k 10 600 rocprofv3 --pmc $counters --output-format csv -d $ROOT/$OUT/pmc_$name -- python3 $ROOT/bench.py --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-side --transform-streams 1 --no-graphs --coder-streams 3 < /dev/null > $ROOT/$OUT/pmc_$name.json 2> $ROOT/$OUT/pmc_$name.err
  echo "pmc pass $name ($counters) rc=$?"
done
cd $ROOT
python profiles/make_pmc_summary.py $OUT $OUT/pmc_summary.json > $OUT/pmc_summary.txt; tail -30 $OUT/pmc_summary.txt
python profiles/make_traffic.py $OUT 24 > $OUT/traffic.log 2>&1; tail -5 $OUT/traffic.log; cp profiles/traffic_conv_gemm.json $OUT/traffic_conv_gemm.json
for d in trace trace_onestream trace_bw0.125 trace_bw0.05; do
  find $OUT/$d -name "*kernel_trace.csv" -delete
  s=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -16 "$s"
done
# the counter files are large: keep only the summary
find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT
