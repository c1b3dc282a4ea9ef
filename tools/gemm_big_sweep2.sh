# EfficientSATRN step against the persistent GEMM's routing rule (min GFLOP x min N), GPU box
for n in 128 96 48 24; do for g in 2.0 1.5 1.0; do
  echo -n "MIN_N=$n MIN_GFLOP=$g: "
 SATRN_KNOBS=gemm_big_min_n=$n,gemm_big_min_gflop=$g python3 bench.py --steps 30 --warmup 5 --no-decode --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(r['ms_per_step'], [(f['kernel'],f['ms']) for f in r['roofline']['families'][:2]])"
done; done
