# EfficientSATRN training step against the persistent GEMM's routing threshold (GPU box): bash tools/minflop_sweep.sh
for v in 2.0 1.8 1.5 1.0; do
  echo "== SATRN_KNOBS=gemm_big_min_gflop=$v"
  SATRN_KNOBS=gemm_big_min_gflop=$v python bench.py --no-extras --no-decode --steps 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], [ (k['kernel'],k['ms']) for k in d.get('kernel_breakdown',[])[:3]])"
done
