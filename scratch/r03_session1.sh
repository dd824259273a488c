#!/bin/bash
# Round 3, GPU session 1 (run from the repo root on the GPU box): the GPU suite on the tree with the loud cut-tile hand-off,
# the new bench line, how the step time moves with the number of coder batches in flight, PMC passes for every kernel of the
# step, and the stdout of the round-2 probes (DESIGN.md section 4 quotes them; they were never kept).
OUT=gpurun_out/r03_a
mkdir -p $OUT
ROOT=$(pwd)
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 600 $OUT/bench.err
# depth sweep: coder batches in flight x latent entropy, default mode and the one-stream mode; and without any coder
for mode in "" "--transform-streams 1 --no-graphs"; do
  tag=$( [ -z "$mode" ] && echo default || echo onestream )
  EAE_BENCH_NO_CODER=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-side $mode 2>/dev/null | python scratch/r03_line.py nocoder $tag >> $OUT/depth_sweep.txt
  for bw in 1.0 0.125 0.05; do
    for n in 2 3 4 5 6 8; do
      timeout 300 python bench.py --steps 30 --warmup 5 --bin-width $bw --coder-streams $n --no-cpu-baseline --no-side $mode 2>/dev/null \
        | python scratch/r03_line.py "bw=$bw n=$n" $tag >> $OUT/depth_sweep.txt
    done
  done
done
cat $OUT/depth_sweep.txt
# the probes
for p in probe_sustain probe_mix probe_coissue probe_mfma4; do
  timeout 180 ./scratch/$p > $OUT/$p.log 2>&1; echo "$p rc=$?"
done
# PMC passes (counters only, program directly behind --): one-stream launch-by-launch mode, 3 steps
cd /tmp && export TMPDIR=/tmp
i=0
for counters in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
                "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $counters --output-format csv -d $ROOT/$OUT/pmc_$i -- python3 $ROOT/bench.py --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-side --transform-streams 1 --no-graphs > $ROOT/$OUT/pmc_$i.json 2> $ROOT/$OUT/pmc_$i.err
  echo "pmc pass $i ($counters) rc=$?"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-side > $ROOT/$OUT/bench_under_rocprof.json 2> $ROOT/$OUT/trace.err
cd $ROOT
python profiles/make_pmc_summary.py $OUT $OUT/pmc_summary.json | tail -40
find $OUT/trace -name "*kernel_trace.csv" -delete
s=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); head -30 "$s"
du -sh $OUT
