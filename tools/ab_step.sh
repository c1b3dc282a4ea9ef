# EfficientSATRN training step, A/B over one SATRN_OFF feature (GPU box): bash tools/ab_step.sh fused_pool_se
sw=$1
for rep in 1 2; do
  python bench.py --no-extras --no-decode --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default      ', d['ms_per_step'], d['final_loss'])"
  env SATRN_OFF=$sw python bench.py --no-extras --no-decode --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SATRN_OFF=$sw', d['ms_per_step'], d['final_loss'])"
done
