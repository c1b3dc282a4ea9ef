# EfficientSATRN training step timed N times, optionally under environment settings (GPU box):
#   bash tools/ab_bench.sh LABEL N [VAR=VALUE ...]   -> "LABEL ms_per_step final_loss" per run
label=$1; n=$2; shift 2
for i in $(seq 1 $n); do
  env "$@" python bench.py --no-extras --no-decode --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['ms_per_step'], d['final_loss'])"
done
