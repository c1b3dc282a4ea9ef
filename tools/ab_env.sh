# A/B of environment variants of the training step on ONE box, alternating runs (GPU box):  bash tools/ab_env.sh "" "VAR=1" "VAR=2 OTHER=1" ...
# prints ms/step (mean, median) per variant and round
for round in 1 2; do
  for v in "$@"; do
    r=$(env $v python3 bench.py --no-extras --no-cpu-baseline --no-decode --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'])")
    echo "round $round [${v:-default}] $r"
  done
done
