#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run from the repo root: bash scratch/collect_profiles.sh <tag>).
# Kernel trace + stats of the default bench command, then the PMC passes (separate runs, counters only).
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-single-image > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; counters=${pass#*:}
  rocprofv3 --pmc $counters --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-single-image > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
cd $ROOT
find $OUT -name "*kernel_stats.csv" | head -3
find $OUT -name "*counter_collection.csv" | head -5
f=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); head -3 "$f"
s=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); head -30 "$s"
# keep the merge small: the per-dispatch traces are large
find $OUT/trace -name "*kernel_trace.csv" -delete
ls -la $OUT $OUT/*/ 2>/dev/null | head -40
du -sh $OUT
