#!/bin/bash
# stream layout sweep: which side streams (creation order) carry the transforms / the coder (EAE_STREAM_LAYOUT)
OUT=gpurun_out/r03_f; mkdir -p $OUT
for bw in 1.0 0.125; do
  for lay in "" "0,1;2,3,6" "0,1;2,3,4" "3,4;0,1,2" "0,1;2,3" "0,4;1,2,5"; do
    EAE_STREAM_LAYOUT="$lay" timeout 300 python bench.py --steps 40 --warmup 8 --bin-width $bw --no-cpu-baseline --no-side 2>/dev/null \
      | python scratch/r03_line.py "bw=$bw 2T layout=[$lay]" | tee -a $OUT/layouts.txt
  done
  for lay in "" ";0,1,2" ";0,1,2,3" ";0,1,2,4" ";0,1,2,4,5"; do
    n=$(echo "$lay" | tr -cd ',' | wc -c); n=$((n+1)); [ -z "$lay" ] && n=3
    EAE_STREAM_LAYOUT="$lay" timeout 300 python bench.py --steps 40 --warmup 8 --bin-width $bw --coder-streams $n --transform-streams 1 --no-graphs --no-cpu-baseline --no-side 2>/dev/null \
      | python scratch/r03_line.py "bw=$bw 1T n=$n layout=[$lay]" | tee -a $OUT/layouts.txt
  done
done
