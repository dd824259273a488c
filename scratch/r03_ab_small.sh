for i in 1 2; do
  for tree in . build/_old_tree; do
    (cd $tree; timeout 300 python bench.py --height 256 --width 256 --batch 64 --steps 60 --warmup 10 --coder-streams 3 --no-cpu-baseline --no-side 2>/dev/null | python /root/repo/scratch/r03_line.py "64x256x256 tree=$tree" | cut -c1-170)
  done
done
