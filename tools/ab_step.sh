# EfficientSATRN training step, A/B over one environment switch (GPU box): bash tools/ab_step.sh SATRN_NO_FUSED_POOL_SE
sw=$1
for rep in 1 2; do
  python bench.py --no-extras --no-decode --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default      ', d['ms_per_step'], d['final_loss'])"
  env $sw=1 python bench.py --no-extras --no-decode --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=1', d['ms_per_step'], d['final_loss'])"
done
