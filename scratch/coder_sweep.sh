#!/bin/bash
# bench.py at three latent entropies x batches of coder work in flight (run on the GPU box: bash scratch/coder_sweep.sh)
for bw in 1.0 0.25 0.125; do
  for n in 2 3 4; do
    timeout 120 python bench.py --steps 20 --warmup 5 --bin-width $bw --coder-streams $n --no-cpu-baseline --no-single-image 2>/dev/null \
      | python scratch/coder_sweep_line.py $n
  done
done
